#!/bin/bash
# round 6, GPU call 16: is the ID / BERT-tiny step host-bound?  eager vs hipGraph replay on the same box
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
B="python bench.py --batch 128 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  $B --tower id 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('id   eager', d['ms_per_step'])"
  $B --tower id --graph 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('id   graph', d['ms_per_step'], d['config']['launch'][:60])"
  $B --bert tiny 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiny eager', d['ms_per_step'])"
  $B --bert tiny --graph 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiny graph', d['ms_per_step'], d['config']['launch'][:60])"
done > $O/g16_graph.txt 2>&1
cat $O/g16_graph.txt
python scripts/host_profile.py id 2>&1 | tail -45 > $O/g16_host.txt; tail -45 $O/g16_host.txt
