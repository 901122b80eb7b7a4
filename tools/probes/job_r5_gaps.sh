R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pg; mkdir -p $R/gpurun_out/r5_gaps
for st in dmd diffusion; do
STAGE=$st CYCLES=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg_$st -o step -- python $R/tools/prof_stage.py > /tmp/pg_$st.log 2>&1
python $R/tools/trace_gaps.py $(ls /tmp/pg_$st/*/step_kernel_trace.csv /tmp/pg_$st/step_kernel_trace.csv 2>/dev/null | head -1) | tee $R/gpurun_out/r5_gaps/$st.txt
done
