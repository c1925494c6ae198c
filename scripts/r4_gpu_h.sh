#!/bin/bash
mkdir -p gpurun_out/r4h
python -m pytest tests/test_fp16_mode_gpu.py::test_deferred_update_equals_the_inline_one tests/test_eval_gpu.py -q > gpurun_out/r4h/t.log 2>&1; tail -3 gpurun_out/r4h/t.log
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r4h/bench_graph.json 2> gpurun_out/r4h/bench_graph.err; grep "timed region\|graph mode\|Error\|error" gpurun_out/r4h/bench_graph.err | head
python bench.py --no-secondary --no-cpu-baseline --no-graph > gpurun_out/r4h/bench_eager.json 2> gpurun_out/r4h/bench_eager.err; grep "timed region" gpurun_out/r4h/bench_eager.err
python bench.py --dtype bf16 --no-secondary --no-cpu-baseline > gpurun_out/r4h/bench_bf16_graph.json 2> gpurun_out/r4h/bench_bf16_graph.err; grep "timed region\|Error" gpurun_out/r4h/bench_bf16_graph.err
for t in "--tower id --batch 128" "--bert tiny --batch 128" "--tower swin_tiny --batch 64 --steps 6 --warmup 2"; do
  python bench.py $t --no-secondary --no-cpu-baseline > gpurun_out/r4h/b.json 2> gpurun_out/r4h/b.err; echo "$t graph: $(grep 'timed region' gpurun_out/r4h/b.err) $(grep -i 'error' gpurun_out/r4h/b.err | head -2)"
  python bench.py $t --no-secondary --no-cpu-baseline --no-graph > gpurun_out/r4h/b.json 2> gpurun_out/r4h/b.err; echo "$t eager: $(grep 'timed region' gpurun_out/r4h/b.err)"
done
