#!/bin/bash
# second set of passes: the VALU instruction mix of the two attention kernels (the first set showed no LDS bank conflict and the LDS array 14-21 % busy)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6/pmc_attn
mkdir -p $O
PASS_D="SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"
PASS_E="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INST_LEVEL_LDS SQ_INSTS_VSKIPPED SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F16"
for prog in attn_pmc prefill_attn_pmc ${EXTRA_PROGS:-}; do
  for p in D E; do
    eval "CTR=\$PASS_$p"
    d=$O/${prog}_$p
    rm -rf $d
    timeout 300 rocprofv3 --pmc $CTR --output-format csv -d $d -- python3 $R/scripts/$prog.py > $O/${prog}_$p.log 2>&1
    echo "$prog pass $p rc $?" >> $O/status.txt
    python3 $R/scripts/pmc_kernel.py $d $O/${prog}_$p.json attn > /dev/null 2>&1
    rm -rf $d
  done
done
cat $O/status.txt
