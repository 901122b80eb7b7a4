cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6m
timeout 600 python -m pytest tests/test_gpu_gemm_sk.py -q 2>&1 | tail -8 | tee gpurun_out/r6m/tests.txt
timeout 600 python tools/bench_gemm.py --sk --shapes dit16,vit16 2>&1 | tee gpurun_out/r6m/gemm_sk.txt | tail -14
