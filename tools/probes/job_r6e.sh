cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6e
timeout 1200 python -m pytest tests/test_gpu_sampler.py::test_diffusion_trainer_vs_reference_capture_c4 tests/test_gpu_dit.py::test_lightningdit_train_route_vs_reference_capture_by_the_bf16_site_oracle tests/test_gpu_fullsize.py::test_diffusion_stage_full_size_c4 -x -q -s 2>&1 | grep -v Warning | tail -40 | tee gpurun_out/r6e/tests.txt
