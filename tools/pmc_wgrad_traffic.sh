# HBM traffic of the weight-gradient kernel (256 x 256 tile, plain 3x3 / 1x1 form) inside the real step: one counter per --pmc pass, dispatches filtered by name.
# usage: bash tools/pmc_wgrad_traffic.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "wgrad_pp_kernel<2, 2, 2, 4, false, false, false, 32>" --output-format csv -d $OUT/$C -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
cd $R && python - $OUT <<'PY' | tee $OUT/traffic_summary.txt
import collections, csv, sys
out = sys.argv[1]
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{c}/t_counter_collection.csv")):
        if "wgrad_pp_kernel" not in r["Kernel_Name"]: continue
        v = agg.setdefault(c, [0.0, 0]); v[0] += float(r["Counter_Value"]); v[1] += 1
rd = 2.0 * 1024.0 * agg["FETCH_SIZE"][0] / agg["FETCH_SIZE"][1]; wr = 1024.0 * agg["WRITE_SIZE"][0] / agg["WRITE_SIZE"][1]
print(f"wgrad_pp_kernel<2, 2, 2, 4, false, false, false, 32>: launches {agg['FETCH_SIZE'][1]}, HBM read {rd/1e6:.1f} MB (FETCH_SIZE x2, gfx950), write {wr/1e6:.1f} MB, total {(rd+wr)/1e6:.1f} MB per launch")
PY
