# HBM traffic of the four shortcut-in-GroupNorm kernels inside the real step (one counter per --pmc pass, kernel-trace only, dispatches filtered by name)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4_short_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
RX='short_apply_kernel|short_bwd_apply_kernel|coop_apply_kernel|coop_bwd_apply_kernel'
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $OUT/$C -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
cd $R && python - <<'PY' | tee $OUT/traffic_summary.txt
import collections, csv, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4_short_pmc")
agg = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{c}/t_counter_collection.csv")):
        n = r["Kernel_Name"]
        k = [s for s in ("short_apply_kernel", "short_bwd_apply_kernel", "coop_apply_kernel", "coop_bwd_apply_kernel") if s in n][0]
        v = agg.setdefault(k, {}).setdefault(c, [0.0, 0]); v[0] += float(r["Counter_Value"]); v[1] += 1
print("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, --kernel-trace only) over python bench.py --steps 2 --warmup 1 --no-cpu-baseline; KiB units;")
print("# FETCH_SIZE doubled (gfx950 reports half the bytes of 16 B/lane coalesced reads, MI355X_MICROARCH.md 'HBM'), WRITE_SIZE as reported")
alg = {"short_apply_kernel": 2684.4, "short_bwd_apply_kernel": 3758.1, "coop_apply_kernel": 1342.2, "coop_bwd_apply_kernel": 1879.0}
for k, d in agg.items():
    rd = 2.0 * 1.024e-3 * d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1]; wr = 1.024e-3 * d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1]
    print(f"{k}: launches {d['FETCH_SIZE'][1]}, HBM read {rd:.1f} MB, write {wr:.1f} MB, total {rd + wr:.1f} MB per launch; algorithmic {alg[k]:.1f} MB -> {(rd + wr) / alg[k]:.2f} x")
PY
rm -rf $OUT/FETCH_SIZE/*kernel_trace* $OUT/WRITE_SIZE/*kernel_trace*
