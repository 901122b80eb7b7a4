cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests/test_gpu_gemm_pp.py tests/test_gpu_dit_stack.py tests/test_gpu_dit.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6j/tests.txt
for rep in 1 2; do
for cfg in "DMVAE_SWIGLU_IN_W12=1" "DMVAE_SWIGLU_IN_W12=0"; do
  echo "== $cfg" | tee -a gpurun_out/r6j/ab.txt
  env $cfg timeout 400 python tools/bench_dmd_step.py 2>&1 | grep -E "student_only|vae_turn" | tee -a gpurun_out/r6j/ab.txt
  env $cfg timeout 400 python tools/bench_diffusion_step.py 2>&1 | grep -E "ms" | tail -2 | tee -a gpurun_out/r6j/ab.txt
done; done
