cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/suite
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/suite/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/suite/bench.log 2>&1; tail -1 gpurun_out/suite/bench.log | cut -c1-200
