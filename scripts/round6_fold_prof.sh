#!/bin/bash
# per-kernel times of a 13-row and a 64-row decode step with RoPE + split folded into the attention kernel (CR_DECODE_FOLD_ROPE=1, default) and as its own launch (=0)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6/fold
mkdir -p $O
for rows in 13 64; do
  for f in 0 1; do
    d=$O/rows${rows}_fold$f
    rm -rf $d
    CR_DECODE_FOLD_ROPE=$f ROWS=$rows STEPS=32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/scripts/decode_rows.py child > $O/rows${rows}_fold$f.log 2>&1
    find $d -name "*kernel_stats.csv" -exec cp {} $O/rows${rows}_fold${f}_kernel_stats.csv \;
    rm -rf $d
  done
done
