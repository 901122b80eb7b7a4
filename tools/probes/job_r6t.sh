cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6t
timeout 600 python -m pytest tests/test_gpu_attention_bwd.py tests/test_gpu_dit.py tests/test_gpu_dit_stack.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r6t/tests.txt
for i in 1 2; do
  for t in 1 0; do
    echo "== DMVAE_ATTN_BWD_TIGHT=$t run $i" | tee -a gpurun_out/r6t/ab.txt
    DMVAE_ATTN_BWD_TIGHT=$t timeout 300 python tools/bench_attention.py 2>&1 | grep "DiT heads" | tee -a gpurun_out/r6t/ab.txt
  done
done
