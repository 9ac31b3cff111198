#!/usr/bin/env python3
"""rocprofv3 --pmc CSV (one pass of the 8 SQ counters of scripts/kernels_pmc.py) -> JSON with the derived fractions per kernel.
usage: pmc_json.py <dir> <out.json>"""
import csv, glob, json, os, sys, collections
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if any(s in k for s in ('attn_kernel', 'gemm256_kernel', 'gemm_skinny')):
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {'_how': 'rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS '
               '--output-format csv -- python3 scripts/kernels_pmc.py (one pass; mean over the launches of each kernel; scripts/pmc_json.py)',
       '_derived': 'mfma_busy_frac = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32); wave_wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / '
                   's_barrier); issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; issuing_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES'}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m['launches'] = len(next(iter(cs.values())))
    if m.get('SQ_BUSY_CYCLES') and m.get('SQ_WAVE_CYCLES'):
        m['mfma_busy_frac'] = round((m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024) / (m['SQ_BUSY_CYCLES'] / 32), 3)
        m['wave_wait_frac'] = round(m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES'], 3)
        m['issue_stall_frac'] = round(m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES'], 3)
        m['issuing_frac'] = round(m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES'], 3)
    res[k[:110]] = m
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk.endswith('_frac')} for k, v in res.items() if isinstance(v, dict)}, indent=1))
