# Round-5 evidence job, one box: the full GPU suite, rocprofv3 kernel trace + stats of the bench command (C2 headline), per-stage traces of the DMD cycle / the
# latent-diffusion step / the adversarial step cut into steps and families, the default bench (all keys), smoke().   usage: bash tools/probes/job_r5_end.sh [tag]
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r5_end}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_prof.log 2>&1
T=$(ls $OUT/prof/*/step_kernel_trace.csv $OUT/prof/step_kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/trace_summary.py $T > $OUT/step_trace_summary.txt 2>&1; rm -f $T
for st in dmd diffusion gan; do
  STAGE=$st CYCLES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$st -o $st -- python $R/tools/prof_stage.py > $OUT/prof_$st.log 2>&1
  T=$(ls $OUT/prof_$st/*/${st}_kernel_trace.csv $OUT/prof_$st/${st}_kernel_trace.csv 2>/dev/null | head -1)
  python $R/tools/stage_trace_summary.py $T 45 > $OUT/${st}_trace_summary.txt 2>&1; rm -f $T
done
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
timeout 600 python tools/bench_dmd_step.py > $OUT/dmd_step.txt 2>&1
timeout 600 python tools/bench_dit.py > $OUT/dit_fwd.txt 2>&1
ONLY=hip timeout 600 python tools/bench_diffusion_step.py > $OUT/diffusion_step.txt 2>&1
timeout 600 python tools/bench_gan_step.py > $OUT/gan_step.txt 2>&1
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/bench.log | cut -c1-300; tail -2 $OUT/smoke.log; tail -3 $OUT/dmd_step.txt; tail -2 $OUT/dit_fwd.txt; tail -1 $OUT/diffusion_step.txt | cut -c1-120; tail -1 $OUT/gan_step.txt | cut -c1-100
