#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
: > $O/summary.txt
run() { name=$1; tmo=$2; shift 2; timeout $tmo "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/summary.txt; }
PT="python -m pytest -q --tb=short -m gpu -p no:cacheprovider -s"
run midsize 400 $PT tests/test_model_gpu.py -k "midsize"
run fp16mode 700 $PT tests/test_fp16_mode_gpu.py
run bench_res32 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-secondary --dtype fp16_res32
cat $O/summary.txt
for f in midsize fp16mode; do echo "=== $f"; grep -E "passed|failed|error|Error|assert|bench-config|g6 |scaler" $O/$f.log | tail -25; done
grep -o '"value": [0-9.]*, "unit"\|"ms_per_step": [0-9.]*, "higher\|"gemm_ms_per_step": [0-9.]*' $O/bench_res32.log | head -4
