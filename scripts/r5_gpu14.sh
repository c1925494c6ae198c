#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
timeout 600 python -m pytest tests/test_swin_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for r in 0 8 4 16 0 8; do
  echo "== rounds $r"; MOREC_SWIN_ATTN_ROUNDS=$r timeout 200 python scripts/swin_attn_bench.py 704 2>&1 | grep -v amdgpu.ids
done
