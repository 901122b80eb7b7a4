cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/suite
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/suite/gpu_suite.txt
