cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6l
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s 2>&1 | grep -E "full size|passed|failed|GiB" | tee gpurun_out/r6l/tests.txt
