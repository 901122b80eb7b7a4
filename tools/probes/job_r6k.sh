cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6k
timeout 900 python -m pytest tests/test_gpu_toy.py tests/test_gpu_dit.py -x -q -s 2>&1 | grep -v Warning | tail -40 | tee gpurun_out/r6k/tests.txt
