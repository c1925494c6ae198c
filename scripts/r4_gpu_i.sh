#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4i
timeout 200 python bench.py --no-secondary --no-cpu-baseline --graph > gpurun_out/r4i/bench_graph.json 2> gpurun_out/r4i/bench_graph.err; echo "rc $?"; grep "timed region\|graph mode\|fault\|Error" gpurun_out/r4i/bench_graph.err | head
timeout 200 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r4i/bench_eager.json 2> gpurun_out/r4i/bench_eager.err; echo "rc $?"; grep "timed region" gpurun_out/r4i/bench_eager.err
