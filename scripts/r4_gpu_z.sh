#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4z
for i in 1 2 3; do
for ts in 0 1; do
MOREC_GEMM8P_TAIL_SPLIT=$ts timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region" | sed "s/^/tail_split=$ts /"
done
done | tee gpurun_out/r4z/tail_split_ab.txt
