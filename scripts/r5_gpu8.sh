#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p $O
for l in hip exp1 exp2 hip exp1 exp2; do
  MOREC_HIP_LIB=$PWD/idvs/morec_amd/libmorec_$l.so timeout 300 python scripts/shadow_bench.py >> $O/shadow.log 2>&1; echo "$l rc=$?"
done
cat $O/shadow.log
