#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel-name x grid, mean counter values over dispatches."""
import collections, csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        if 'conv' not in r['Kernel_Name'] and 'wgrad' not in r['Kernel_Name'] and 'kl_' not in r['Kernel_Name']:
            continue
        key = (r['Kernel_Name'][:40], r['Grid_Size'])
        d = agg.setdefault(key, collections.OrderedDict())
        v = d.setdefault(r['Counter_Name'], [0.0, 0]); v[0] += float(r['Counter_Value']); v[1] += 1
    for key, d in agg.items():
        print(key, {k: round(v[0] / v[1]) for k, v in d.items()})
