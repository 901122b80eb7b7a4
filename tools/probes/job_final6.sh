cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_final
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r6_final/gpu_suite.txt
timeout 600 python bench.py 2>gpurun_out/r6_final/bench.err | tee gpurun_out/r6_final/bench_line.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/r6_final/smoke.txt
