#!/usr/bin/env python
"""counter_collection CSVs of tools/pmc_traffic_r3.sh (one rocprofv3 --pmc pass per counter) -> mean HBM read / write bytes per launch for each profiled kernel,
printed and written as the JSON files bench.py reads back (profiles/rN_conv_pp_traffic.json, profiles/rN_wgrad_pp_traffic.json).
Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads -- what these kernels issue -- so reads are doubled; WRITE_SIZE as reported."""
import collections, csv, json, sys
out, rnd = sys.argv[1], int(sys.argv[2])
agg = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{c}/t_counter_collection.csv")):
        name = r["Kernel_Name"]
        short = name[name.index("conv_pp_kernel") if "conv_pp_kernel" in name else name.index("wgrad_pp_kernel"):].split("(")[0].rstrip(">") + ">"
        d = agg.setdefault(short, {}).setdefault(int(r["Grid_Size"]), {})
        v = d.setdefault(c, [0.0, 0]); v[0] += float(r["Counter_Value"]); v[1] += 1
CORR = "FETCH_SIZE, WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of 16 B/lane coalesced reads, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as reported"
CMD = "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --kernel-include-regex <kernels> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (tools/pmc_traffic_r3.sh, one counter per pass)"
res = {}
for k, grids in agg.items():
    tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
    per = {}
    for g, d in sorted(grids.items()):
        for c in tot:
            tot[c][0] += d[c][0]; tot[c][1] += d[c][1]
        per["grid %d" % g] = {"launches": d["FETCH_SIZE"][1], "read_MB": round(2.0 * 1.024e-3 * d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1], 1),
                              "write_MB": round(1.024e-3 * d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1], 1)}
    rd = 2.0 * 1.024e-3 * tot["FETCH_SIZE"][0] / tot["FETCH_SIZE"][1]
    wr = 1.024e-3 * tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1]
    res[k] = {"kernel": k, "command": CMD, "launches_profiled": tot["FETCH_SIZE"][1], "hbm_read_MB_per_launch": round(rd, 1), "hbm_write_MB_per_launch": round(wr, 1),
              "hbm_MB_per_launch": round(rd + wr, 1), "per_grid": per, "corrections": CORR, "round": rnd}
    print(f"{k}: launches {tot['FETCH_SIZE'][1]}, HBM read {rd:.1f} MB, write {wr:.1f} MB, total {rd + wr:.1f} MB per launch; per grid {per}")
conv = [v for k, v in res.items() if k.startswith("conv_pp")]
wg = [v for k, v in res.items() if k.startswith("wgrad_pp")]
if conv:
    json.dump(conv[0], open(f"{out}/conv_pp_traffic.json", "w"), indent=1)
if wg:
    main = [v for v in wg if "<2, 2, 2, 4, false, false, false, 32>" in v["kernel"]]
    top = dict(main[0] if main else wg[0])
    top["kernels"] = wg
    json.dump(top, open(f"{out}/wgrad_pp_traffic.json", "w"), indent=1)
