#!/usr/bin/env python
"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace CSV (steps delimited by the optimiser kernel)."""
import collections
import csv
import sys


def main(path, nsteps=3, marker="adamw_ema_kernel", top=60):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    nsteps = min(nsteps, len(ends) - 1)
    sel = rows[ends[-nsteps - 1] + 1: ends[-1] + 1]
    wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / nsteps / 1e6
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[r["Kernel_Name"]]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    tot = sum(a[0] for a in agg.values()) / nsteps / 1e6
    print(f"# steady state over the last {nsteps} steps: wall {wall:.2f} ms/step, sum of kernel durations {tot:.2f} ms/step, "
          f"{len(sel) / nsteps:.0f} kernels/step")
    print("#  ms/step  calls/step     avg_us  kernel")
    for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{d / nsteps / 1e6:9.3f} {c / nsteps:10.1f} {d / c / 1e3:11.1f}  {k[:140]}")


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:3]))
