R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_gan2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gan.py tests/test_gpu_modules.py tests/test_gpu_train_step.py -q --tb=short -x 2>&1 | tail -6
timeout 600 python tools/bench_gan_step.py 2>&1 | tail -2 | cut -c1-300 | tee $OUT/gan_step.txt
