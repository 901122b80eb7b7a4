R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_timeline; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1; cd $R
T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python tools/trace_timeline.py $T > $OUT/timeline.txt 2>&1; rm -f $T; head -3 $OUT/timeline.txt
