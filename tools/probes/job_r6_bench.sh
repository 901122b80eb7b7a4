cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_bench
timeout 900 python bench.py 2>gpurun_out/r6_bench/bench.err | tee gpurun_out/r6_bench/bench_line.json | cut -c1-400
