R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5_ditfwd; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ONLY=hip timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o dit -- python $R/tools/bench_dit.py > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/root/repo/gpurun_out/r5_ditfwd/p/**/dit_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>6}  avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:120]}')
PY
tail -3 $OUT/log.txt
