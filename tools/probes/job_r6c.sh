cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6c
timeout 600 python -m pytest tests/test_gpu_gemm_sk.py -q 2>&1 | tail -25 | tee gpurun_out/r6c/tests.txt
timeout 600 python tools/bench_gemm.py --sk --shapes dit16,vit16,"vit fc2","vit fc1","vit qkv","dit64 w3","dit64 d_" 2>&1 | tee gpurun_out/r6c/gemm_sk.txt | tail -25
