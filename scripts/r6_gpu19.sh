#!/bin/bash
# round 6, GPU call 19: slab fold with batched loads -- weight-gradient tests, kernel averages in the Swin-T / BERT-tiny / text steps
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_shapes_gpu.py tests/test_deterministic_gpu.py -x -q 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof; mkdir -p /tmp/prof
for spec in "sw --tower swin_tiny --batch 64 --steps 4 --warmup 2" "ti --bert tiny --batch 128 --steps 20 --warmup 5" "tx --steps 4 --warmup 2"; do
  set -- $spec; tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o $tag -- python $R/bench.py $@ --no-cpu-baseline --no-secondary > $O/g19_${tag}_line.json 2>/dev/null
  python $R/scripts/prof_summary.py /tmp/prof/${tag}_results.db 1 "$tag" | grep -E "reduce_slabs|total kernel" 
  tail -1 $O/g19_${tag}_line.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"
done 2>&1 | tee $O/g19_slabs.txt
