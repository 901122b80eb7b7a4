cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6v
timeout 900 python -m pytest tests/test_gpu_attention_bwd.py tests/test_gpu_dit.py tests/test_gpu_dit_stack.py tests/test_gpu_sampler.py -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r6v/tests.txt
for i in 1 2; do
  for t in 1 0; do
    echo "== DMVAE_QK_UNPADDED=$t run $i" | tee -a gpurun_out/r6v/ab.txt
    ONLY=hip STEPS=20 DMVAE_QK_UNPADDED=$t timeout 600 python tools/bench_diffusion_step.py 2>&1 | grep -v Warn | tail -2 | tee -a gpurun_out/r6v/ab.txt
  done
done
