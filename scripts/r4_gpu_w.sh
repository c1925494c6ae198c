#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4w
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "gemm_tn" 2>&1 | tail -5 | tee gpurun_out/r4w/tn_tests.txt
MOREC_TN8P_SKIP=0 timeout 200 python scripts/tn_skip_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4w/tn_noskip.txt
timeout 200 python scripts/tn_skip_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4w/tn_skip.txt
for i in 1 2; do
timeout 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
MOREC_TN8P_SKIP=0 timeout 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
done
timeout 300 python bench.py --tower swin_base --batch 32 --steps 4 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
MOREC_TN8P_SKIP=0 timeout 300 python bench.py --tower swin_base --batch 32 --steps 4 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
timeout 900 python -m pytest tests/test_swin_gpu.py tests/test_bench_mode_parity_vision_gpu.py -x -q 2>&1 | tail -3
