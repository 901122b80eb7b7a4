R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_gaps; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --time-every 100 > $OUT/bench_prof.log 2>&1; cd $R
T=$(ls $OUT/prof/*/step_kernel_trace.csv $OUT/prof/step_kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_gaps.py $T > $OUT/gaps.txt 2>&1; rm -f $T; head -45 $OUT/gaps.txt
