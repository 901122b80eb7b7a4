R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_c; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_sampler.py "tests/test_gpu_fullsize.py::test_dmd_stage_full_size_cycle_c3" -x -q -s 2>&1 | grep -v "Warning\|warnings.warn\|^$\|lp = LPIPS" | tail -30 > $OUT/pytest.log
tail -12 $OUT/pytest.log
for i in 1 2; do
  DMVAE_ALLOW_STOCK=1 DMVAE_TMP_STOCK_ADALN=1 timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | sed 's/^/stock-adaLN: /'
  timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | sed 's/^/rows-kernel: /'
done | tee $OUT/ab_dmd.log
