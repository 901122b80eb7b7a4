R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_grouped; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_dit_stack.py tests/test_gpu_dit.py tests/test_gpu_vit_train.py tests/test_gpu_kernels.py tests/test_gpu_conv_c2_shapes.py -q --tb=short -x 2>&1 | tail -15
timeout 1200 python -m pytest tests -m gpu -x -q -k "dmd or diffusion or trainer or sampler or step" 2>&1 | tail -4
timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -4 | tee $OUT/dmd_step.txt
ONLY=hip timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-200 | tee $OUT/diffusion_step.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-1500 | tee $OUT/bench_c2.txt
