R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_overlap_trace; mkdir -p $OUT; export TMPDIR=/tmp
export DMVAE_WGRAD_STREAM=1 DMVAE_WGRAD_CUS=192
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
cd $R
T=$(ls $OUT/prof/*/step_kernel_trace.csv $OUT/prof/step_kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_timeline.py $T > $OUT/timeline.txt 2>&1
head -1 $T > $OUT/trace_header.txt
rm -f $T
head -3 $OUT/timeline.txt
