R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_b; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_linear_rows.py tests/test_gpu_dit.py tests/test_gpu_train_step.py tests/test_gpu_sampler.py "tests/test_gpu_fullsize.py::test_dmd_stage_full_size_cycle_c3" -x -q -s 2>&1 | grep -v "Warning\|warnings.warn\|^$\|lp = LPIPS" | tail -30 > $OUT/pytest.log
tail -22 $OUT/pytest.log
timeout 600 python tools/bench_dit.py 2>&1 | tail -4 | tee $OUT/bench_dit.log
timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -4 | tee $OUT/bench_dmd.log
