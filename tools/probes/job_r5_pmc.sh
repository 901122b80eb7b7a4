# Round-5 PMC traffic: the C2 step's three MFMA kernels (bench.py reads the resulting JSON back), and the DMD stage's new kernels; then the stage traces again for the family tables
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_pmc; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/pmc_traffic_r3.sh r5_pmc 5 2>&1 | tail -6
bash tools/pmc_stage_traffic.sh r5_pmc dmd 'wgrad_pp_grouped_kernel|attention_bwd_lse_kernel|rms_gate_bwd_kernel|rows_wgrad_mfma_kernel|splitk_sum_kernel|adamw_ema_kernel' 2>&1 | tail -14
cd /tmp
for st in dmd diffusion gan; do
  STAGE=$st CYCLES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$st -o $st -- python $R/tools/prof_stage.py > $OUT/prof_$st.log 2>&1
  T=$(ls $OUT/prof_$st/*/${st}_kernel_trace.csv $OUT/prof_$st/${st}_kernel_trace.csv 2>/dev/null | head -1)
  python $R/tools/stage_trace_summary.py $T 45 > $OUT/${st}_trace_summary.txt 2>&1; rm -f $T
done
head -14 $OUT/dmd_trace_summary.txt
