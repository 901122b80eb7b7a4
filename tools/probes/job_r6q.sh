cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6q
for st in 0 8000 16000 30000 0 16000 45000; do
  echo "== DMVAE_ATTN_STAGGER=$st" | tee -a gpurun_out/r6q/ab.txt
  DMVAE_ATTN_STAGGER=$st timeout 300 python tools/bench_attention.py 2>&1 | grep "DiT heads B= 64" | tee -a gpurun_out/r6q/ab.txt
done
