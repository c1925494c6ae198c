#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4y
timeout 900 python -m pytest tests/test_swin_gpu.py -q -x -k "fp16 or window_attention or patchify_merge" 2>&1 | tail -4 | tee gpurun_out/r4y/swin_fp16_tests.txt
timeout 600 python -m pytest tests/test_swin_gpu.py -q -s -k "fp16 and full_size" 2>&1 | grep -E "item-vector|passed|failed" | tee -a gpurun_out/r4y/swin_fp16_tests.txt
timeout 600 python -m pytest tests/test_bench_mode_parity_vision_gpu.py -q -s -k fp16 2>&1 | grep -E "vision fp16|scaler|passed|failed|assert|Error" | cut -c1-400 | tee gpurun_out/r4y/vision_fp16_parity.txt
for dt in fp16 bf16; do
timeout 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 3 --dtype $dt --no-secondary --no-cpu-baseline 2> gpurun_out/r4y/bench_t_$dt.err > gpurun_out/r4y/bench_t_$dt.json; grep "timed region" gpurun_out/r4y/bench_t_$dt.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r4y/bench_t_$dt.json') if l.startswith('{')][-1]); print('$dt', d['ms_per_step'], d['dtype'], d.get('loss_scaler_state'), d.get('final_loss'))"
done
timeout 300 python bench.py --tower swin_base --batch 32 --steps 4 --warmup 3 --dtype fp16 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
