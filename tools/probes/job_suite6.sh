cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_suite
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r6_suite/gpu_suite.txt
