#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python scripts/small_gemm_check.py f16 > $O/g14_small.txt 2>&1
python scripts/small_gemm_check.py bf16 >> $O/g14_small.txt 2>&1
cat $O/g14_small.txt | grep -v amdgpu.ids
python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_shapes_gpu.py -x -q 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --tower id --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('id', d['ms_per_step'])"; done
for i in 1 2; do python bench.py --bert tiny --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiny', d['ms_per_step'])"; done
