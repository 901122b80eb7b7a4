"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs of tools/bench_attention.py -> per (attention kernel, grid): mean HBM bytes per launch
(units / gfx950 correction as tools/pmc_traffic_summary.py: KiB, FETCH_SIZE doubled)."""
import collections, csv, sys
agg = collections.OrderedDict()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "attention" not in name:
            continue
        key = (name.split("(")[0][-64:], int(r["Grid_Size"]))
        v = agg.setdefault(key, {}).setdefault(r["Counter_Name"], [0.0, 0])
        v[0] += float(r["Counter_Value"]); v[1] += 1
for (k, grid), d in agg.items():
    rd = 2.0 * 1024.0 * d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1] if "FETCH_SIZE" in d else float("nan")
    wr = 1024.0 * d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1] if "WRITE_SIZE" in d else float("nan")
    n = d.get("FETCH_SIZE", d.get("WRITE_SIZE"))[1]
    print(f"{k} grid={grid}: launches {n}, HBM read {rd/1e6:.1f} MB, write {wr/1e6:.1f} MB per launch")
