cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6g
timeout 1200 python -m pytest tests/test_gpu_parity_fp32.py -q -k "lightningdit or diffusion_step or patchgan" 2>&1 | grep -v Warning | tail -60 | tee gpurun_out/r6g/tests.txt
