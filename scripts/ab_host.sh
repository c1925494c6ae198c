#!/bin/bash
# host-side launch cost A/B on the launch-bound lines: raw stream handle (default) vs a torch.cuda.Stream object per launch
ulimit -c 0
for i in 1 2; do
for v in 0 1; do
echo "MOREC_STREAM_OBJ=$v"
MOREC_STREAM_OBJ=$v timeout 200 python bench.py --tower id --batch 128 --steps 40 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region" | sed "s/^/  id   /"
MOREC_STREAM_OBJ=$v timeout 200 python bench.py --bert tiny --batch 128 --steps 40 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region" | sed "s/^/  tiny /"
MOREC_STREAM_OBJ=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed region" | sed "s/^/  base /"
done
done
