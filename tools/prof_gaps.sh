R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pg
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o step -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/trace_gaps.py $(ls /tmp/pg/*/step_kernel_trace.csv /tmp/pg/step_kernel_trace.csv 2>/dev/null | head -1)
