export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/kl
rocprofv3 --kernel-trace --output-format csv -d /tmp/kl -o k -- python $GRAFT_REPO_ROOT/tools/probes/klmmd_lat.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = (glob.glob("/tmp/kl/*/k_kernel_trace.csv") + glob.glob("/tmp/kl/k_kernel_trace.csv"))[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "mmd_pair" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"].split("(")[0][-28:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Grid_Size_Y", ""))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    v = sorted(v)[: max(1, len(v) * 3 // 4)]
    print(k, f"{sum(v)/len(v)/1e3:.1f} us (n={len(v)})")
PY
