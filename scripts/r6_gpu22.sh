#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_gemm_small_gpu.py -x -q 2>&1 | tail -5
for i in 1 2 3; do python bench.py --tower id --batch 128 --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('id', d['ms_per_step'])"; python bench.py --bert tiny --batch 128 --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiny', d['ms_per_step'])"; done 2>&1 | tee $O/g22_id_tiny.txt
