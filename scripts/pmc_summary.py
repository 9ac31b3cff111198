#!/usr/bin/env python3
"""Sum rocprofv3 --pmc CSV rows per (kernel, counter): usage pmc_summary.py <dir> [kernel substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if sub in k:
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:32s} n={len(v):3d} last={v[-1]:.4g} mean={sum(v)/len(v):.4g}')
