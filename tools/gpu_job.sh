# usage: bash tools/gpu_job.sh <tag> [pytest] [prof] [bench]
set -x
R=$GRAFT_REPO_ROOT; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for what in "$@"; do
case $what in
pytest) timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log;;
prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $OUT/bench_prof.log 2>&1); python tools/trace_summary.py $OUT/prof/step_kernel_trace.csv 3 > $OUT/trace_summary.txt; head -30 $OUT/trace_summary.txt;;
bench) timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.log 2>&1; tail -2 $OUT/bench.log;;
esac
done
