# profile of the DMD stage + diffusion step after the stack-function batch
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_prof2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -4 | tee $OUT/dmd_step.txt
ONLY=hip timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-200 | tee $OUT/diffusion_step.txt
cd /tmp
for st in dmd diffusion; do
  STAGE=$st CYCLES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$st -o $st -- python $R/tools/prof_stage.py > $OUT/prof_$st.log 2>&1
  T=$(ls $OUT/prof_$st/*/${st}_kernel_trace.csv $OUT/prof_$st/${st}_kernel_trace.csv 2>/dev/null | head -1)
  python $R/tools/stage_trace_summary.py $T 45 > $OUT/${st}_trace_summary.txt 2>&1
  rm -f $T
done
head -60 $OUT/dmd_trace_summary.txt
