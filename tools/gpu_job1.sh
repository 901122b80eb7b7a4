set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r1a
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r1a/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1a/pytest_gpu.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1a/prof -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r1a/bench_prof.log 2>&1
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r1a/bench.log 2>&1
ls -la gpurun_out/r1a gpurun_out/r1a/prof/* | head -40
tail -3 gpurun_out/r1a/pytest_gpu.log; tail -2 gpurun_out/r1a/bench.log
