#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (one pass each) -> per (kernel, grid): mean HBM read / write bytes per
launch.  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": the derived metrics are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide (16 B/lane) coalesced reads -- what these kernels issue -- so reads are doubled; WRITE_SIZE is taken as
reported (uncalibrated in the guide; compare with the output tensor size printed alongside)."""
import collections, csv, sys
agg = collections.OrderedDict()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "conv" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0][-70:], int(r["Grid_Size"]))
        d = agg.setdefault(key, {})
        v = d.setdefault(r["Counter_Name"], [0.0, 0]); v[0] += float(r["Counter_Value"]); v[1] += 1
for (k, grid), d in agg.items():
    rd = 2.0 * 1024.0 * d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1] if "FETCH_SIZE" in d else float("nan")
    wr = 1024.0 * d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1] if "WRITE_SIZE" in d else float("nan")
    print(f"{k} grid={grid}: launches {d.get('FETCH_SIZE', d.get('WRITE_SIZE'))[1]}, HBM read {rd/1e6:.1f} MB, write {wr/1e6:.1f} MB, total {(rd+wr)/1e6:.1f} MB per launch")
