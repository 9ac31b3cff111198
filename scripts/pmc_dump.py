#!/usr/bin/env python3
"""Sum the counters of a rocprofv3 --pmc run per kernel (development aid): python scripts/pmc_dump.py <dir> [substring]"""
import csv, sys, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if len(sys.argv) > 2 and sys.argv[2] not in k: continue
        d[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
for k, v in d.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f'   {c:34s} {x:16.0f}  ({n[(k, c)]} launches)  per launch {x / n[(k, c)]:14.0f}')
