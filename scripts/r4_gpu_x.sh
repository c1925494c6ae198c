#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4x
timeout 300 python scripts/swin_gemm_shapes.py 704 96 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4x/swin_t_shapes.txt
timeout 300 python scripts/swin_gemm_shapes.py 352 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4x/swin_b_shapes.txt
