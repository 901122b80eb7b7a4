# Kernel trace of the bench command + per-kernel summary (no tests).  usage: bash tools/prof_step.sh <tag> [extra env assignments are inherited]
R=$GRAFT_REPO_ROOT; TAG=${1:-prof}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
cd $R
python tools/trace_summary.py $(ls $OUT/prof/*/step_kernel_trace.csv $OUT/prof/step_kernel_trace.csv 2>/dev/null | head -1) > $OUT/trace_summary.txt 2>&1
rm -f $OUT/prof/*/step_kernel_trace.csv $OUT/prof/step_kernel_trace.csv
head -40 $OUT/trace_summary.txt | cut -c1-150
