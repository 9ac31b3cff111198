#!/usr/bin/env python3
"""Per-kernel summary of one decode step out of a rocprofv3 --kernel-trace database (development aid).
   python scripts/trace_summary.py gpurun_out/prof_x/dec_results.db"""
import sqlite3, collections, re, sys, statistics
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
picks = [i for i, r in enumerate(rows) if 'pick_kernel' in r[0]]
a, b = picks[-6], picks[-5]
step = rows[a + 1:b + 1]
print('launches per step', len(step), 'wall us', round((step[-1][2] - step[0][1]) / 1e3, 1), 'sum of kernel us', round(sum(r[2] - r[1] for r in step) / 1e3, 1))
agg = collections.OrderedDict()
for r in step:
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', r[0])
    key = (n[:70], r[3], r[4])
    d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += (r[2] - r[1]) / 1e3
for k, v in agg.items():
    print(f'{v[0]:4d} x {v[1] / v[0]:8.2f} us = {v[1]:8.1f}  grid {k[1]} wg {k[2]}  {k[0]}')
