#!/bin/bash
# development aid: kernel-only times of the decode GEMMs at M rows (rocprofv3 kernel trace), one line per kernel
# usage: scripts/skinny_prof.sh TAG M   (env CR_PARTIAL_RT_SMALL etc. pass through)
TAG=$1; M=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp_$TAG
SLICED=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$TAG -- python3 /root/repo/scripts/skinny_bench.py $M > /tmp/sp_$TAG.log 2>&1
f=$(find /tmp/sp_$TAG -name "*kernel_trace.csv" | head -1)
echo "== $TAG M=$M"
python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_skinny' in r['Kernel_Name']:
        d[(r['Kernel_Name'].split('gemm_skinny_kernel')[1][:22], r['Grid_Size_X'], r['Grid_Size_Y'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in d.items():
    v.sort()
    print(f'{v[len(v)//2]/1e3:8.2f} us (min {v[0]/1e3:7.2f}) x{len(v):4d}  {k}')
PY
