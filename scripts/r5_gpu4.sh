#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
: > $O/summary.txt
run() { name=$1; tmo=$2; shift 2; timeout $tmo "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/summary.txt; }
PT="python -m pytest -q --tb=short -m gpu -p no:cacheprovider -s"
run attn 400 $PT tests/test_kernels_gpu.py -k "attention"
run dropout 400 $PT tests/test_dropout_gpu.py
run model 600 $PT tests/test_model_gpu.py -k "g18 or midsize or g19"
run attn_bench 200 python scripts/attn64_bench.py
MOREC_ATTN_MFMA64=0 run attn_bench_valu 200 python scripts/attn64_bench.py
cat $O/summary.txt
for f in attn dropout model; do echo "=== $f"; grep -E "passed|failed|error|Error|assert|g18 " $O/$f.log | tail -12; done
cat $O/attn_bench.log $O/attn_bench_valu.log | grep MFMA64
