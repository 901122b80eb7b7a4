cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_attn2
for i in 1 2 3; do
DMVAE_ATTN_3BUF=0 python tools/bench_attention.py 2>&1 | grep "DiT"
python tools/bench_attention.py 2>&1 | grep "DiT"
done | tee gpurun_out/r5_attn2/bench3.txt
