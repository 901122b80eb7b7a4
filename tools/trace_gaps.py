#!/usr/bin/env python
"""Where the stream sits idle inside a steady-state step: the largest gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV, with the kernels on
either side (steps delimited by the optimiser kernel), and the idle time summed by the kernel that FOLLOWS the gap."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
sel = rows[ends[-2] + 1: ends[-1] + 1]
gaps = []
for a, b in zip(sel, sel[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"].split("(")[0][-50:], b["Kernel_Name"].split("(")[0][-50:]))
tot = sum(g for g, _, _ in gaps if g > 0)
print(f"kernels {len(sel)}, idle {tot/1e6:.2f} ms of wall {(int(sel[-1]['End_Timestamp'])-int(sel[0]['Start_Timestamp']))/1e6:.2f} ms; gaps > 20 us: {sum(1 for g,_,_ in gaps if g > 20000)}")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"{g/1e3:8.1f} us  after {a:50s} before {b}")
by = collections.Counter()
for g, a, b in gaps:
    if g > 0: by[b] += g
print("idle by following kernel:")
for k, v in by.most_common(12):
    print(f"{v/1e3:8.1f} us  {k}")
