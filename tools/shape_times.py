#!/usr/bin/env python
"""Per-launch-shape durations of selected kernels in the last step of a rocprofv3 kernel trace."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'adamw_ema_kernel' in r['Kernel_Name']]
sel = rows[ends[-2] + 1:ends[-1] + 1]
for name in sys.argv[2:]:
    agg = collections.OrderedDict()
    for r in sel:
        if name in r['Kernel_Name']:
            key = (r['Kernel_Name'][:50], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'], r['VGPR_Count'], r['Accum_VGPR_Count'])
            a = agg.setdefault(key, [0, 0]); a[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); a[1] += 1
    for k, (d, c) in agg.items():
        print(f"{d / 1e3:9.1f} us total {c:3d}x avg {d / c / 1e3:8.1f} us", k)
