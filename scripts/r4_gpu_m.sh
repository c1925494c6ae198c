#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4m
T=tests/test_bench_mode_parity_vision_gpu.py
for i in 1 2; do python -m pytest $T -q -s 2>&1 | grep "vision bench-mode\|passed\|failed" | cut -c1-420 > gpurun_out/r4m/head_$i.log; echo "HEAD run $i: $(tail -1 gpurun_out/r4m/head_$i.log)"; done
MOREC_GEMM_SKINNY=1 python -m pytest $T -q -s 2>&1 | grep "vision bench-mode\|passed\|failed" | cut -c1-420 > gpurun_out/r4m/noskinny.log; echo "no skinny: $(tail -1 gpurun_out/r4m/noskinny.log)"
MOREC_HIP_LIB=$PWD/scratch_libs/libmorec_oldattn.so python -m pytest $T -q -s 2>&1 | grep "vision bench-mode\|passed\|failed" | cut -c1-420 > gpurun_out/r4m/oldattn.log; echo "old attn: $(tail -1 gpurun_out/r4m/oldattn.log)"
MOREC_GEMM_SKINNY=1 MOREC_HIP_LIB=$PWD/scratch_libs/libmorec_oldattn.so python -m pytest $T -q -s 2>&1 | grep "vision bench-mode\|passed\|failed" | cut -c1-420 > gpurun_out/r4m/both_off.log; echo "both off: $(tail -1 gpurun_out/r4m/both_off.log)"
cat gpurun_out/r4m/*.log | grep "vision bench-mode" | cut -c1-400
