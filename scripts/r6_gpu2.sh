#!/bin/bash
# round 6, GPU call 2: -amdgpu-mfma-vgpr-form A/B (attention kernels), deterministic Swin, g21 yardstick
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_deterministic_gpu.py tests/test_swin_gpu.py tests/test_kernels_gpu.py tests/test_eval_gpu.py -x -q > $O/g2_tests.txt 2>&1
tail -4 $O/g2_tests.txt
grep -E "g13|g15" $O/g2_tests.txt | head
{
for rep in 1 2; do
for L in scratch_libs/libmorec_base.so idvs/morec_amd/libmorec_hip.so; do
  echo "== $L"
  MOREC_HIP_LIB=$PWD/$L python scripts/swin_attn_bench.py 704 2>&1 | grep -v amdgpu.ids
  MOREC_HIP_LIB=$PWD/$L python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
  MOREC_HIP_LIB=$PWD/$L python scripts/attn64_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
done
done
} > $O/g2_attn_ab.txt 2>&1
B="python bench.py --no-secondary --no-cpu-baseline"
for rep in 1 2; do
for L in scratch_libs/libmorec_base.so idvs/morec_amd/libmorec_hip.so; do
  MOREC_HIP_LIB=$PWD/$L $B --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L fp16', d['ms_per_step'])"
  MOREC_HIP_LIB=$PWD/$L $B --tower swin_tiny --batch 64 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L swin_tiny', d['ms_per_step'])"
  MOREC_HIP_LIB=$PWD/$L $B --tower swin_base --batch 32 --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L swin_base', d['ms_per_step'])"
done
done > $O/g2_step_ab.txt 2>&1
cat $O/g2_step_ab.txt
