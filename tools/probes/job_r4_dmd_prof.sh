R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_dmd_prof; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o dmd -- python $R/tools/bench_dmd_step.py > $OUT/bench.log 2>&1; cd $R
tail -4 $OUT/bench.log
T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1); rm -f $T
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY' | tee $OUT/top.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%8.2f ms %6d calls %9.1f us  %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:110]))
PY
