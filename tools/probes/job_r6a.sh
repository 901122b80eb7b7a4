cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_reparam.py tests/test_gpu_dist.py tests/test_gpu_dit_stack.py tests/test_gpu_train_step.py tests/test_gpu_toy.py -x -q 2>&1 | tail -15 | tee gpurun_out/r6a/tests.txt
timeout 600 python tools/bench_gemm.py --shapes dit16,vit16 2>&1 | tee gpurun_out/r6a/gemm_base.txt | tail -12
