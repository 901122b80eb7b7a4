R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_norms; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dit.py tests/test_gpu_dit_stack.py -q -x --tb=short 2>&1 | tail -3
echo "== eight channels per lane (default)" | tee $OUT/norms.txt; timeout 300 python tools/bench_dit_norms.py 2>&1 | grep "B=" | tee -a $OUT/norms.txt
echo "== DMVAE_RM8=0" | tee -a $OUT/norms.txt; DMVAE_RM8=0 timeout 300 python tools/bench_dit_norms.py 2>&1 | grep "B=" | tee -a $OUT/norms.txt
