cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5_attn_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --kernel-include-regex 'attention_bwd_lse_kernel|attention_kernel' --output-format csv -d $OUT/$N -o t -- python $R/tools/bench_attention.py > $OUT/$N.log 2>&1
  echo "$N rc=$?"
done
cd $R; python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.OrderedDict()
for f in glob.glob("gpurun_out/r5_attn_pmc/*/**/t_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-48:], int(r["Grid_Size"]), r["Counter_Name"])
        v = agg.setdefault(k, [0.0, 0]); v[0] += float(r["Counter_Value"]); v[1] += 1
last = None
for (k, g, c), v in sorted(agg.items()):
    if (k, g) != last:
        print(f"## {k} grid {g}"); last = (k, g)
    print(f"   {c:34s} {v[0] / v[1]:16.0f}")
PY
find $OUT -name "*kernel_trace*" -delete
