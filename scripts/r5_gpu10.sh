#!/bin/bash
# Swin window attention: launch times per stage, WIDE on/off for the backward, and SQ counters of the stage-1 launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5l
mkdir -p $O
timeout 200 python scripts/swin_attn_bench.py 704 > $O/wide1.log 2>&1; echo "rc=$?"
MOREC_SWIN_BWD_WIDE=0 timeout 200 python scripts/swin_attn_bench.py 704 > $O/wide0.log 2>&1; echo "rc=$?"
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf /tmp/pp
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pp -o sw -- python $R/scripts/swin_attn_bench.py 176 > $O/pmc_run.log 2>&1
  echo "# --pmc $set (n_img 176)" >> $O/pmc.txt
  python $R/scripts/pmc_summary.py /tmp/pp/sw_results.db "%swin_attn%" >> $O/pmc.txt 2>&1
done
cat $O/wide1.log $O/wide0.log | grep -v amdgpu.ids
