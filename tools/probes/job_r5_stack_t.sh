R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_gemm; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dit_stack.py tests/test_gpu_dit.py tests/test_gpu_vit_train.py tests/test_gpu_gemm_pp.py -q --tb=line 2>&1 | tail -12
timeout 900 python tools/bench_gemm.py --sweep --cold --shapes "dit16,vit16" 2>&1 | tee $OUT/sweep_cold.txt | cut -c1-400 | tail -20
