#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4q
timeout 200 python scripts/race_probe2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4q/race2.txt
timeout 300 python scripts/swin_nan_probe.py 10 2>&1 | grep -v amdgpu | cut -c1-300 | tee gpurun_out/r4q/nan.txt | tail -3
python -m pytest tests/test_swin_gpu.py tests/test_kernels_gpu.py -q -k "attention or swin" > gpurun_out/r4q/t.log 2>&1; tail -2 gpurun_out/r4q/t.log
python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
python bench.py --tower swin_base --batch 32 --steps 4 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
