#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5o
mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  rm -rf /tmp/pp
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pp -o sw -- python $R/scripts/swin_attn_bench.py 704 > $O/pmc_run.log 2>&1
  echo "# --pmc $c (n_img 704; per-dispatch averages over the 7 stage/shift shapes x 11 calls; FETCH/WRITE_SIZE in KiB)" >> $O/pmc.txt
  python $R/scripts/pmc_summary.py /tmp/pp/sw_results.db "%swin_attn%" >> $O/pmc.txt 2>&1
done
cat $O/pmc.txt
