#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
run() { name=$1; tmo=$2; shift 2; timeout $tmo "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
B="python bench.py --no-cpu-baseline --no-secondary"
MOREC_GEMM8P_TMR=0 run st0 300 $B --tower swin_tiny --batch 64 --steps 6 --warmup 2
run st1 300 $B --tower swin_tiny --batch 64 --steps 6 --warmup 2
MOREC_GEMM8P_TMR=0 run sb0 300 $B --tower swin_base --batch 32 --steps 4 --warmup 2
run sb1 300 $B --tower swin_base --batch 32 --steps 4 --warmup 2
MOREC_GEMM8P_TMR=0 run st0b 300 $B --tower swin_tiny --batch 64 --steps 6 --warmup 2
run st1b 300 $B --tower swin_tiny --batch 64 --steps 6 --warmup 2
run id 300 $B --tower id --batch 128 --steps 20 --warmup 5
run tiny 300 $B --bert tiny --batch 128 --steps 20 --warmup 5
for f in st0 st1 sb0 sb1 st0b st1b id tiny; do echo "== $f: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/$f.log | head -1) $(grep -o '"gemm_ms_per_step": [0-9.]*' $O/$f.log | head -1) $(grep -o '"frac": [0-9.]*' $O/$f.log | head -1)"; done
