#!/bin/bash
# round-5 GPU call 1: new tests (res32 LayerNorm, deterministic mode, multi-attribute fused step, DDP drop-in, run.py under torchrun) + quick bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
: > $O/summary.txt
run() { name=$1; tmo=$2; shift 2; timeout $tmo "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/summary.txt; }
PT="python -m pytest -q --tb=short -m gpu -p no:cacheprovider -s"
run ln_res32 300 $PT tests/test_layernorm_res32_gpu.py
run determ 400 $PT tests/test_deterministic_gpu.py
run g19_midsize 400 $PT tests/test_model_gpu.py -k "g19 or midsize"
run fp16mode 700 $PT tests/test_fp16_mode_gpu.py
run ddp 900 $PT tests/test_ddp_dropin_gpu.py
run graph 400 $PT tests/test_graph_step_gpu.py
run driver 500 $PT tests/test_eval_gpu.py -k run_driver
run bench_fp16 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-secondary
run bench_res32 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-secondary --dtype fp16_res32
cat $O/summary.txt
for f in ln_res32 determ g19_midsize fp16mode ddp graph driver; do echo "=== $f"; grep -E "passed|failed|error|Error|assert|bench-config|g6 |g19 |DDP drop-in|deterministic x|graph vs eager" $O/$f.log | tail -25; done
tail -c 1500 $O/bench_fp16.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -4
tail -c 3000 $O/bench_res32.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -4
