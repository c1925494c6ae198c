#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
: > $O/summary.txt
run() { name=$1; tmo=$2; shift 2; timeout $tmo "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/summary.txt; }
PT="python -m pytest -q --tb=short -m gpu -p no:cacheprovider -s"
run tmr_test 600 $PT tests/test_gemm_bench_shapes_gpu.py -k "tile_height or gemm8p_absolute"
run tmr_bench 400 python scripts/tmr_bench.py
MOREC_GEMM8P_TMR=0 run bench_tmr0 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary
run bench_tmr1 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary
MOREC_GEMM8P_TMR=0 run bench_tmr0b 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary
run bench_tmr1b 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary
cat $O/summary.txt
grep -E "passed|failed|Error|assert" $O/tmr_test.log | tail -8
cat $O/tmr_bench.log | grep "M="
for f in bench_tmr0 bench_tmr1 bench_tmr0b bench_tmr1b; do echo "== $f: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/$f.log | head -1) $(grep -o '"gemm_ms_per_step": [0-9.]*' $O/$f.log | head -1) $(grep -o '"frac": [0-9.]*' $O/$f.log | head -1)"; done
