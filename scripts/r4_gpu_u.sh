#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4u
timeout 400 python -m pytest tests/test_gemm_skinny_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/r4u/skinny_tests.txt
timeout 200 python scripts/mlp_recompute_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4u/mlp_bench.txt
timeout 900 python -m pytest tests/test_swin_gpu.py tests/test_bench_mode_parity_vision_gpu.py -x -q > gpurun_out/r4u/swin_tests.txt 2>&1; tail -3 gpurun_out/r4u/swin_tests.txt
for i in 1 2; do
timeout 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
MOREC_GEMM_SKINNY=2 timeout 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
done
