#!/bin/bash
# round 6, GPU call 15: ID / BERT-tiny step A/B of gemm_small + the ln_bwd row floor (same box, alternating), traces afterwards
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT; mkdir -p $O
python scripts/small_gemm_check.py f16 2>&1 | grep -v amdgpu.ids > $O/g15_small.txt
B="python bench.py --batch 128 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary"
for rep in 1 2 3; do
  for m in 1 0; do
    MOREC_GEMM_SMALL=$m $B --tower id 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('id   gemm_small_mode=$m', d['ms_per_step'])"
    MOREC_GEMM_SMALL=$m $B --bert tiny 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiny gemm_small_mode=$m', d['ms_per_step'])"
  done
done > $O/g15_ab.txt 2>&1
cat $O/g15_ab.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof; mkdir -p /tmp/prof
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o id -- python $R/bench.py --tower id --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/g15_id_line.json 2>/dev/null
python $R/scripts/trace_step_full.py /tmp/prof/id_results.db gather_rows_kernel 12 > $O/g15_id_step.txt 2>&1
python $R/scripts/prof_summary.py /tmp/prof/id_results.db 34 "id tower" > $O/g15_id_stats.csv 2>&1
tail -2 $O/g15_id_step.txt
