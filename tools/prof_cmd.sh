# Kernel trace + per-kernel summary of an arbitrary python command.  usage: bash tools/prof_cmd.sh <tag> <python args...>
R=$GRAFT_REPO_ROOT; TAG=$1; shift; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python "$@" > $OUT/run.log 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = (glob.glob(out + "/prof/*/t_kernel_stats.csv") + glob.glob(out + "/prof/t_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%  {int(r["Calls"]):6d} x {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
rm -f $OUT/prof/*/t_kernel_trace.csv $OUT/prof/t_kernel_trace.csv
tail -2 $OUT/run.log
