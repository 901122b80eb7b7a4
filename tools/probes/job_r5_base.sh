# Round 5 baseline of the DMD stage (C3) and the latent-diffusion step (C4): wall-clock benches + rocprofv3 kernel trace cut into steps and families
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_base; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/bench_dmd_step.py > $OUT/dmd_step.txt 2>&1
timeout 600 python tools/bench_dit.py > $OUT/dit_fwd.txt 2>&1
ONLY=hip timeout 600 python tools/bench_diffusion_step.py > $OUT/diffusion_step.txt 2>&1
cd /tmp
for st in dmd diffusion; do
  STAGE=$st timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$st -o $st -- python $R/tools/prof_stage.py > $OUT/prof_$st.log 2>&1
  T=$(ls $OUT/prof_$st/*/${st}_kernel_trace.csv $OUT/prof_$st/${st}_kernel_trace.csv 2>/dev/null | head -1)
  python $R/tools/stage_trace_summary.py $T 40 > $OUT/${st}_trace_summary.txt 2>&1
  rm -f $T
done
cd $R
tail -n 4 $OUT/dmd_step.txt $OUT/dit_fwd.txt $OUT/diffusion_step.txt; head -50 $OUT/dmd_trace_summary.txt
