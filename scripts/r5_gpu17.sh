#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5r
mkdir -p $O
timeout 600 python -m pytest tests/test_swin_gpu.py tests/test_bench_mode_parity_vision_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
echo "== with dbias"; timeout 200 python scripts/swin_attn_bench.py 704 2>&1 | grep stage
echo "== without dbias"; SAB_NODBIAS=1 timeout 200 python scripts/swin_attn_bench.py 704 2>&1 | grep stage
B="python bench.py --no-cpu-baseline --no-secondary"
for m in 1 8 1 8; do
  MOREC_SWIN_WPW_MIN=$m timeout 300 $B --tower swin_tiny --batch 64 --steps 6 --warmup 2 > $O/st_$m.log 2>&1
  echo "swin_tiny wpw_min $m: $(grep -o '"ms_per_step": [0-9.]*' $O/st_$m.log | head -1)"
  MOREC_SWIN_WPW_MIN=$m timeout 300 $B --tower swin_base --batch 32 --steps 4 --warmup 2 > $O/sb_$m.log 2>&1
  echo "swin_base wpw_min $m: $(grep -o '"ms_per_step": [0-9.]*' $O/sb_$m.log | head -1)"
done
