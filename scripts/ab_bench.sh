#!/bin/bash
# A/B of whole-step variants inside ONE GPU call (same box, interleaved repeats): scripts/ab_bench.sh <outdir> <label>=<env assignments> ...
# e.g.  scripts/ab_bench.sh gpurun_out/ab "base=MOREC_WGRAD_STREAM=0" "side=MOREC_WGRAD_STREAM=1"
out=$1; shift
mkdir -p "$out"
for rep in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    env $envs python bench.py --steps ${STEPS:-12} --warmup 4 --no-cpu-baseline --no-secondary ${BENCH_ARGS} > "$out/${label}_$rep.json" 2>> "$out/err.log"
  done
done
python - "$out" <<'PY'
import json, sys, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_[0-9].json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        res[os.path.basename(f).rsplit("_", 1)[0]].append((j["ms_per_step"], j["roofline"]["gemm_ms_per_step"], j["roofline"]["frac"], j.get("final_loss")))
    except Exception as e:
        res[os.path.basename(f)].append(("ERR", str(e)))
for k, v in res.items():
    print(k, v)
PY
