#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4k
python -m pytest tests/test_gemm_skinny_gpu.py tests/test_swin_gpu.py -q > gpurun_out/r4k/t.log 2>&1; tail -3 gpurun_out/r4k/t.log
python scripts/skinny_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4k/skinny_bench.txt
python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/r4k/swin.json 2> gpurun_out/r4k/swin.err; grep "timed region" gpurun_out/r4k/swin.err
python bench.py --tower swin_base --batch 32 --steps 4 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/r4k/swinb.json 2> gpurun_out/r4k/swinb.err; grep "timed region" gpurun_out/r4k/swinb.err
