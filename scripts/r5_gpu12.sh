#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5n
mkdir -p $O
timeout 600 python -m pytest tests/test_swin_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for l in base hip base hip; do
  echo "== $l"; MOREC_HIP_LIB=$PWD/idvs/morec_amd/libmorec_$l.so timeout 200 python scripts/swin_attn_bench.py 704 2>&1 | grep -v amdgpu.ids
done
