R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pg_dmd
STAGE=dmd CYCLES=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg_dmd -o step -- python $R/tools/prof_stage.py > /tmp/pg_dmd.log 2>&1
python $R/tools/trace_gaps.py $(ls /tmp/pg_dmd/*/step_kernel_trace.csv /tmp/pg_dmd/step_kernel_trace.csv 2>/dev/null | head -1) | head -24
