#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4o
echo "== HEAD"; timeout 300 python scripts/swin_nan_probe.py 8 2>&1 | grep -v amdgpu | cut -c1-700 | tee gpurun_out/r4o/head.txt | tail -9
echo "== no skinny"; MOREC_GEMM_SKINNY=1 timeout 300 python scripts/swin_nan_probe.py 8 2>&1 | grep -v amdgpu | cut -c1-400 | tee gpurun_out/r4o/noskinny.txt | tail -2
echo "== old attention"; MOREC_HIP_LIB=$PWD/scratch_libs/libmorec_oldattn.so timeout 300 python scripts/swin_nan_probe.py 8 2>&1 | grep -v amdgpu | cut -c1-400 | tee gpurun_out/r4o/oldattn.txt | tail -2
