#!/bin/bash
# round 6, GPU call 18: kernel stats of the fp32x3 step
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof; mkdir -p /tmp/prof
MOREC_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o x3 -- python $R/bench.py --dtype fp32x3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/g18_x3_line.json 2>/dev/null
NS=$(python -c "import json,sys; print(json.loads([l for l in open('$O/g18_x3_line.json') if l.startswith('{')][-1])['steps_executed'])")
python $R/scripts/prof_summary.py /tmp/prof/x3_results.db $NS "r06 fp32x3: MOREC_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- bench.py --dtype fp32x3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ($NS steps traced; single stream)" > $O/g18_x3_kernel_stats.csv
head -40 $O/g18_x3_kernel_stats.csv | cut -c1-150
