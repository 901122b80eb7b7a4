cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6i
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6i/tests.txt
timeout 900 python bench.py > gpurun_out/r6i/bench.log 2>&1; tail -1 gpurun_out/r6i/bench.log > gpurun_out/r6i/bench_line.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6i/bench_line.json').read())
for k in ("value","ms_per_step","roofline","roofline_wgrad","linear_gemm","comm_n1","c3_dmd_cycle","c4_diffusion_step","gan_step","kl_mmd"):
    print(k, json.dumps(d.get(k))[:600])
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
