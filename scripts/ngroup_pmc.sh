#!/bin/bash
# FETCH_SIZE of the encoder GEMM shapes under the two tile orders of gemm8p (row-major vs column groups): bash scripts/ngroup_pmc.sh <out.txt>
R=$GRAFT_REPO_ROOT; OUT=${1:-$R/gpurun_out/ngroup_pmc.txt}
cd /tmp; export TMPDIR=/tmp; mkdir -p /tmp/prof
{
for ng in 0 -1 4 3; do
  for spec in "54919 2304 768 bias" "54919 3072 768 gelu" "54919 3072 768 dmul" "54919 768 3072 nt" "54919 768 768 nt"; do
    echo "# MOREC_GEMM8P_NGROUP=$ng rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python scripts/gemm_one.py $spec   (KiB per dispatch, to be doubled: MI355X_MICROARCH.md HBM section)"
    rm -f /tmp/prof/ng_results.db
    MOREC_GEMM8P_NGROUP=$ng timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof -o ng -- python $R/scripts/gemm_one.py $spec > /dev/null 2>&1
    python $R/scripts/pmc_summary.py /tmp/prof/ng_results.db "%gemm8p%"
  done
done
} > $OUT 2>&1
