set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_gemm_pp.py -x -q 2>&1 | tail -5
python tools/bench_gemm.py --shapes dit16,dit64 --cold --rounds 5 2>&1 | tail -10
python tools/bench_dit.py 2>&1 | tail -8
