cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6r
DMVAE_SK_LOCAL=1 timeout 600 python -m pytest tests/test_gpu_gemm_sk.py -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r6r/tests_local.txt
for loc in 0 1 0 1; do
  echo "== DMVAE_SK_LOCAL=$loc" | tee -a gpurun_out/r6r/ab.txt
  DMVAE_SK_LOCAL=$loc timeout 600 python tools/bench_gemm.py --sk --cold --shapes "dit16 w3,dit16 d_qkv,dit16 d_w12,vit16 fc2,dit16 w12" 2>&1 | grep "M=" | tee -a gpurun_out/r6r/ab.txt
done
