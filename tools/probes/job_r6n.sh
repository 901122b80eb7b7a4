cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6n
timeout 1500 python -m pytest tests/test_gpu_train_step.py::test_dmd_trainer_vs_reference_capture_c3 tests/test_gpu_parity_fp32.py::test_dmd_stage_steps_vs_reference_f32 -q -s 2>&1 | grep -v Warning | tail -60 | tee gpurun_out/r6n/tests.txt
