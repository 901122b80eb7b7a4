# HBM traffic of named kernels inside a stage (tools/prof_stage.py: dmd | diffusion | gan): one counter per --pmc pass (kernel-trace only), dispatches filtered by name.
# usage: bash tools/pmc_stage_traffic.sh <tag> <stage> '<kernel regex>'   -> gpurun_out/<tag>/stage_traffic_<stage>.txt (per kernel x grid: launches, mean read / write MB;
# FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM")
R=$GRAFT_REPO_ROOT; TAG=$1; ST=$2; RX=$3; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  STAGE=$ST CYCLES=1 timeout 900 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $OUT/${ST}_$C -o t -- python $R/tools/prof_stage.py > $OUT/${ST}_$C.log 2>&1
  echo "$C rc=$?"
done
cd $R && python - "$OUT" "$ST" <<'PY' | tee $OUT/stage_traffic_$ST.txt
import collections, csv, sys
out, st = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{st}_{c}/t_counter_collection.csv")):
        d = agg.setdefault((r["Kernel_Name"].split("(")[0][-90:], int(r["Grid_Size"])), {})
        v = d.setdefault(c, [0.0, 0]); v[0] += float(r["Counter_Value"]); v[1] += 1
print(f"# stage {st}: mean HBM MB per launch (FETCH_SIZE x 2 x 1.024e-3, WRITE_SIZE x 1.024e-3)")
for (k, g), d in agg.items():
    f, w = d.get("FETCH_SIZE", [0, 1]), d.get("WRITE_SIZE", [0, 1])
    print(f"{k:90s} grid {g:8d} launches {f[1]:4d}  read {2.0 * 1.024e-3 * f[0] / max(f[1], 1):9.1f} MB  write {1.024e-3 * w[0] / max(w[1], 1):9.1f} MB")
PY
rm -rf $OUT/${ST}_FETCH_SIZE/*kernel_trace* $OUT/${ST}_WRITE_SIZE/*kernel_trace*
