#!/usr/bin/env python
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV when kernels run on more than one stream: every kernel of the last full step
(steps delimited by the optimiser kernel) as  start (us from step start), duration, queue, how many other kernels overlap it and for how long, name.
usage: python tools/trace_timeline.py <kernel_trace.csv> [first_us] [last_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
sel = rows[ends[-2] + 1: ends[-1] + 1]
t0 = int(sel[0]["Start_Timestamp"])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e12
iv = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r) for r in sel]
qs = {}
busy = 0
cur_end = 0
for s, e, r in iv:
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print(f"# step wall {(iv[-1][1])/1e6:.3f} ms, union of kernel intervals {busy/1e6:.3f} ms, sum of durations {sum(e-s for s,e,_ in iv)/1e6:.3f} ms, {len(iv)} kernels")
for i, (s, e, r) in enumerate(iv):
    if s / 1e3 < lo or s / 1e3 > hi:
        continue
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    ov = sum(max(0, min(e, e2) - max(s, s2)) for j, (s2, e2, _) in enumerate(iv) if j != i and s2 < e and e2 > s)
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmvae_", "")[:70]
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} q{q} ov {ov/1e3:7.1f}  {name}")
