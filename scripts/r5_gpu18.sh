#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
cd /tmp
rm -rf /tmp/prof; mkdir -p /tmp/prof
MOREC_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o kb -- python $R/bench.py --tower swin_base --batch 32 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $O/r05_prof_swin_base_stats_line.json 2> /dev/null
NS=$(python -c "import json,sys; print(json.loads([l for l in open('$O/r05_prof_swin_base_stats_line.json') if l.startswith('{')][-1])['steps_executed'])")
python $R/scripts/prof_summary.py /tmp/prof/kb_results.db $NS "r05 swin_base B=32 (352 images/step): MOREC_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- bench.py --tower swin_base --batch 32 --steps 4 --warmup 2 ($NS steps traced)" > $O/r05_swin_base_kernel_stats.csv
head -40 $O/r05_swin_base_kernel_stats.csv | cut -c1-150
