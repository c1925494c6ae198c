#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4id
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/profid; mkdir -p /tmp/profid
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/profid -o kt -- python $R/bench.py --tower id --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/r4id/line.json 2>/dev/null
NS=$(python -c "import json; print(json.loads([l for l in open('$R/gpurun_out/r4id/line.json') if l.startswith('{')][-1])['steps_executed'])")
python $R/scripts/prof_summary.py /tmp/profid/kt_results.db $NS "r04 id tower: rocprofv3 --kernel-trace --stats -- bench.py --tower id --batch 128 --steps 20 --warmup 5 ($NS steps traced, two streams)" > $R/gpurun_out/r4id/id_kernel_stats.csv
head -60 $R/gpurun_out/r4id/id_kernel_stats.csv | cut -c1-150
echo "steps $NS"
