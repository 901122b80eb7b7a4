R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_final2; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1; cd $R
T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python tools/trace_summary.py $T > $OUT/trace_summary.txt 2>&1
python tools/shape_times.py $T > $OUT/step_shapes.txt 2>&1 || true
rm -f $T
head -3 $OUT/trace_summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench_line.json; cut -c1-400 $OUT/bench_line.json
timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3
timeout 600 python tools/bench_gan_step.py 2>&1 | tail -2
