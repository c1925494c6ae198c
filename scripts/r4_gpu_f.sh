#!/bin/bash
mkdir -p gpurun_out/r4f
python -m pytest tests/test_fp16_mode_gpu.py -q -s -k "deferred" > gpurun_out/r4f/defer_test.log 2>&1; tail -3 gpurun_out/r4f/defer_test.log
python -m pytest tests/test_swin_gpu.py tests/test_bench_mode_parity_vision_gpu.py -q > gpurun_out/r4f/swin_test.log 2>&1; tail -3 gpurun_out/r4f/swin_test.log
for d in 1 0; do MOREC_DEFER_UPDATE=$d python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r4f/bench_defer$d.json 2> gpurun_out/r4f/bench_defer$d.err; grep "timed region" gpurun_out/r4f/bench_defer$d.err; done
for s in 0 1; do MOREC_GEMM_SKINNY=$s python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/r4f/swin_sk$s.json 2> gpurun_out/r4f/swin_sk$s.err; grep "timed region" gpurun_out/r4f/swin_sk$s.err; done
MOREC_GEMM_SKINNY=0 python bench.py --tower swin_base --batch 32 --steps 4 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/r4f/swinb.json 2> gpurun_out/r4f/swinb.err; grep "timed region" gpurun_out/r4f/swinb.err
