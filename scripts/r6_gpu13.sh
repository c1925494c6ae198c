#!/bin/bash
# round 6, GPU call 13: kernel-by-kernel trace of one ID-tower step and one BERT-tiny step (launch order, grids)
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof; mkdir -p /tmp/prof
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o id -- python $R/bench.py --tower id --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/g13_id_line.json 2>/dev/null
python $R/scripts/trace_step_full.py /tmp/prof/id_results.db gather_rows_kernel 12 > $O/g13_id_step.txt 2>&1
python $R/scripts/prof_summary.py /tmp/prof/id_results.db 34 "id tower" > $O/g13_id_stats.csv 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ti -- python $R/bench.py --bert tiny --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/g13_tiny_line.json 2>/dev/null
python $R/scripts/trace_step_full.py /tmp/prof/ti_results.db bert_embed_fwd_kernel 12 > $O/g13_tiny_step.txt 2>&1
python $R/scripts/prof_summary.py /tmp/prof/ti_results.db 34 "bert tiny" > $O/g13_tiny_stats.csv 2>&1
tail -3 $O/g13_id_step.txt; tail -3 $O/g13_tiny_step.txt
