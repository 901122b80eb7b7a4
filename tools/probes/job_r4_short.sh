cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4_short2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_norm_short.py -x -q 2>&1 | tail -15
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $(find $OUT/prof -name '*kernel_trace.csv' | head -1) > $OUT/trace_summary.txt 2>&1; find $OUT/prof -name '*kernel_trace.csv' -delete
grep -n "steady\|short\|coop\|apply_kernel\|bwd_partial" $OUT/trace_summary.txt | cut -c1-170
for r in 1 2 3; do for v in 0 1; do
  timeout 600 python tools/probes/ab_flag.py dmvae_amd.functional SHORTCUT_IN_NORM_512 $v --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('coop512=$v', d['ms_per_step'], d['ms_per_step_windows'], d['env']['sclk_mhz_avg'])" | tee -a $OUT/ab.txt
done; done
