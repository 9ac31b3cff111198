#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs -> per-kernel means for kernels whose name contains one of the given substrings (development aid).
usage: pmc_kernel.py <dir> <out.json> substr [substr ...]"""
import csv, glob, json, os, sys, collections
d, out, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if any(s in k for s in subs):
            acc[k[:90]][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, cs in acc.items():
    res[k] = {c: {'mean': sum(v) / len(v), 'launches': len(v)} for c, v in cs.items()}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1)[:3000])
