cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4_wt; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_conv_thin.py tests/test_gpu_norm_conv_out.py tests/test_gpu_modules.py -x -q 2>&1 | tail -3
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $(find $OUT/prof -name '*kernel_trace.csv' | head -1) > $OUT/trace_summary.txt 2>&1; find $OUT/prof -name '*kernel_trace.csv' -delete
grep -n "steady\|wgrad_thin\|conv_thin_kernel<64" $OUT/trace_summary.txt | cut -c1-150
