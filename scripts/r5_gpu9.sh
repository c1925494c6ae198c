#!/bin/bash
# permlane-swap butterflies + dead padded-key registers in the attention kernels: parity tests, then A/B against the previous build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
timeout 900 python -m pytest tests/test_swin_gpu.py tests/test_kernels_gpu.py tests/test_dropout_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python bench.py --no-cpu-baseline --no-secondary"
for l in base hip base hip; do
  MOREC_HIP_LIB=$PWD/idvs/morec_amd/libmorec_$l.so timeout 300 $B --tower swin_tiny --batch 64 --steps 6 --warmup 2 > $O/st_$l.log 2>&1
  echo "swin_tiny $l: $(grep -o '"ms_per_step": [0-9.]*' $O/st_$l.log | head -1)"
done
for l in base hip base hip; do
  MOREC_HIP_LIB=$PWD/idvs/morec_amd/libmorec_$l.so timeout 300 $B --steps 8 --warmup 3 > $O/tx_$l.log 2>&1
  echo "text $l: $(grep -o '"ms_per_step": [0-9.]*' $O/tx_$l.log | head -1)"
done
for l in base hip; do
  echo "attn64 $l:"; MOREC_HIP_LIB=$PWD/idvs/morec_amd/libmorec_$l.so timeout 120 python scripts/attn64_bench.py 2>&1 | grep -v amdgpu.ids
done
