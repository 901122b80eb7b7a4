R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_end; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^E *frame\|^frame\|Warning\|warnings.warn\|^  *lp = LPIPS\|^tests/" | tail -40 > $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1; cd $R
T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python tools/trace_summary.py $T > $OUT/trace_summary.txt 2>&1; rm -f $T
head -1 $OUT/trace_summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench_line.json; cut -c1-330 $OUT/bench_line.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -1
timeout 600 python tools/bench_gan_step.py 2>&1 | tail -1 | cut -c1-80
timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-120
timeout 600 python tools/bench_vit_train.py 2>&1 | tail -1 | cut -c1-120
