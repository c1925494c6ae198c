#!/bin/bash
# round 6, GPU call 17: fp32 attention on the matrix cores -- tests, per-launch times, fp32x3 / fp32 step A/B
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -5
python -m pytest tests/test_dropout_gpu.py tests/test_fp32x3_gpu.py -x -q 2>&1 | tail -3
{ MOREC_ATTN_F32_MFMA=0 python scripts/attn_f32_bench.py; python scripts/attn_f32_bench.py; } 2>&1 | grep -v amdgpu.ids > $O/g17_attn_f32.txt
cat $O/g17_attn_f32.txt
python -m pytest tests/test_model_gpu.py tests/test_train_step_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary"
for m in 0 1; do
  MOREC_ATTN_F32_MFMA=$m $B --dtype fp32x3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 mfma_attn=$m', d['ms_per_step'], d['final_loss'])"
  MOREC_ATTN_F32_MFMA=$m $B --dtype fp32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32   mfma_attn=$m', d['ms_per_step'], d['final_loss'])"
done > $O/g17_step_ab.txt 2>&1
cat $O/g17_step_ab.txt
