cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4_lp; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_conv_in3.py tests/test_gpu_lpips.py tests/test_gpu_train_step.py tests/test_gpu_norm_conv_out.py -x -q 2>&1 | tail -4
for r in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; done
