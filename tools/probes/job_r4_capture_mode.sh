cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_dit.py -x -q 2>&1 | tail -1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kl_mmd']['fused_B32'])"
