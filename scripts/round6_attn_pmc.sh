#!/bin/bash
# LDS / matrix-pipe counters of the two attention kernels (vit_attn_kernel, flash_attn_kernel<128, causal>).
# Separate --pmc passes (8 SQ slots per pass); the program itself after `--`.  Output: gpurun_out/r6/pmc_attn/*.json
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6/pmc_attn
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*" $O/counters_list.txt | sort -u > $O/counters_lds_mfma.txt
PASS_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
PASS_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
PASS_C="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
for prog in attn_pmc prefill_attn_pmc; do
  for p in A B C; do
    eval "CTR=\$PASS_$p"
    d=$O/${prog}_$p
    rm -rf $d
    timeout 300 rocprofv3 --pmc $CTR --output-format csv -d $d -- python3 $R/scripts/$prog.py > $O/${prog}_$p.log 2>&1
    echo "$prog pass $p rc $?" >> $O/status.txt
  done
  python3 $R/scripts/pmc_kernel.py $O/${prog}_A $O/${prog}_A.json attn > /dev/null 2>&1
  python3 $R/scripts/pmc_kernel.py $O/${prog}_B $O/${prog}_B.json attn > /dev/null 2>&1
  python3 $R/scripts/pmc_kernel.py $O/${prog}_C $O/${prog}_C.json attn > /dev/null 2>&1
  # timing of the same launches without counters
  d=$O/${prog}_trace
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/scripts/$prog.py > $O/${prog}_trace.log 2>&1
  find $d -name "*kernel_stats.csv" -exec cp {} $O/${prog}_kernel_stats.csv \;
done
# keep only the summaries (the raw per-dispatch CSVs are small here, but the .db files are not)
find $O -name "*.db" -delete
cat $O/status.txt
