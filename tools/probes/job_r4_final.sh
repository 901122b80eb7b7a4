R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^E *frame\|^frame\|Warning\|warnings.warn\|^  *lp = LPIPS\|^tests/" | tail -40 > $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1; cd $R
T=$(ls $OUT/prof/*/step_kernel_trace.csv $OUT/prof/step_kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_timeline.py $T > $OUT/timeline.txt 2>&1; rm -f $T
grep "conv_pp_kernel<128, 512" $OUT/timeline.txt | head -30
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
