# Round-end GPU job: full -m gpu suite, rocprofv3 kernel trace + stats of the bench command, a plain bench run, smoke().  usage: bash tools/gpu_job.sh <tag>
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-job}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
cd $R
python tools/trace_summary.py $(ls $OUT/prof/*/step_kernel_trace.csv $OUT/prof/step_kernel_trace.csv 2>/dev/null | head -1) > $OUT/trace_summary.txt 2>&1
rm -f $OUT/prof/*/step_kernel_trace.csv $OUT/prof/step_kernel_trace.csv      # tens of MB; the stats CSV and the summary are what is kept
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/bench.log | cut -c1-400; tail -2 $OUT/smoke.log; head -12 $OUT/trace_summary.txt
