cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6o
timeout 900 python -m pytest tests/test_gpu_parity_fp32.py -q -k "gan or patchgan" 2>&1 | grep -v Warning | tail -40 | tee gpurun_out/r6o/tests.txt
