R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_inf; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_dit_stack.py tests/test_gpu_dit.py tests/test_gpu_vit_train.py -q --tb=short -x 2>&1 | tail -6
timeout 1200 python -m pytest tests -m gpu -x -q -k "dmd or diffusion or trainer or sampler or step" 2>&1 | tail -3
timeout 600 python tools/bench_dit.py 2>&1 | tail -2 | tee $OUT/dit_fwd.txt
timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -4 | tee $OUT/dmd_step.txt
ONLY=hip timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-200 | tee $OUT/diffusion_step.txt
