cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6p
timeout 600 python -m pytest tests/test_gpu_attention_bwd.py tests/test_gpu_dit.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r6p/tests.txt
for i in 1 2; do
  echo "== DMVAE_ATTN_PIPE=1 (default) run $i" | tee -a gpurun_out/r6p/ab.txt
  timeout 300 python tools/bench_attention.py 2>&1 | grep "DiT heads" | tee -a gpurun_out/r6p/ab.txt
  echo "== DMVAE_ATTN_PIPE=0 run $i" | tee -a gpurun_out/r6p/ab.txt
  DMVAE_ATTN_PIPE=0 timeout 300 python tools/bench_attention.py 2>&1 | grep "DiT heads" | tee -a gpurun_out/r6p/ab.txt
done
