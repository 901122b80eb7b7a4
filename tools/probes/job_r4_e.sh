R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_e; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_linear_rows.py tests/test_gpu_dit.py tests/test_gpu_train_step.py tests/test_gpu_sampler.py tests/test_gpu_vit_train.py "tests/test_gpu_fullsize.py::test_dmd_stage_full_size_cycle_c3" -x -q -s 2>&1 | grep -v "Warning\|warnings.warn\|^$\|lp = LPIPS\|^tests/" | tail -24 > $OUT/pytest.log
tail -14 $OUT/pytest.log
