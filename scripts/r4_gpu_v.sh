#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4v
for i in 1 2 3; do
timeout 600 python -m pytest tests/test_fp16_mode_gpu.py::test_deferred_update_equals_the_inline_one tests/test_graph_step_gpu.py -q -s 2>&1 | grep -E "graph vs eager|deferred vs inline|passed|failed" | cut -c1-300
done | tee gpurun_out/r4v/noise.txt
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r4v/gputests.txt 2>&1; tail -6 gpurun_out/r4v/gputests.txt
