#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
: > $O/summary.txt
run() { name=$1; tmo=$2; shift 2; timeout $tmo "$@" > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/summary.txt; }
PT="python -m pytest -q --tb=short -m gpu -p no:cacheprovider -s"
run images 300 $PT tests/test_images.py
run driver 500 $PT tests/test_eval_gpu.py -k run_driver
run fp16mode 700 $PT tests/test_fp16_mode_gpu.py -k g6
run swin_res 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary
run swin_u8_feed 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --vision-input u8
MOREC_BENCH_INLINE_INPUT=1 run swin_u8_inline 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --vision-input u8
run swin_u8_feed2 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --vision-input u8
run swin_res2 300 python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary
cat $O/summary.txt
for f in images driver fp16mode; do echo "=== $f"; grep -E "passed|failed|error|Error|assert|g6 " $O/$f.log | tail -12; done
for f in swin_res swin_u8_feed swin_u8_inline swin_u8_feed2 swin_res2; do echo "== $f: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/$f.log | head -1) $(grep -o '"host_pack_ms_per_batch": [0-9.]*' $O/$f.log | head -1)"; done
