cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6d
timeout 300 python -m pytest tests/test_gpu_gemm_sk.py tests/test_gpu_dit_stack.py -q 2>&1 | tail -5 | tee gpurun_out/r6d/tests.txt
for rep in 1 2; do
for cfg in "DMVAE_SPLITK_FUSED=1 DMVAE_SPLITK=3" "DMVAE_SPLITK_FUSED=0 DMVAE_SPLITK=3" "DMVAE_SPLITK=0" "DMVAE_SPLITK_FUSED=1 DMVAE_SPLITK=2"; do
  echo "== $cfg" | tee -a gpurun_out/r6d/ab.txt
  env $cfg timeout 400 python tools/bench_dmd_step.py 2>&1 | grep -E "ms/step" | tee -a gpurun_out/r6d/ab.txt
done; done
