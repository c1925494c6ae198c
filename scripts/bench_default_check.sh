#!/bin/bash
# the driver's bench command, end to end, with a wall clock and a digest of the line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r5i}
mkdir -p $O
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "rc=$? wall=$(( $(date +%s) - T0 ))s"
python - "$O/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dtype")}, "frac", d["roofline"]["frac"], "gemm ms", d["roofline"]["gemm_ms_per_step"], "traffic/alg", d["roofline"].get("traffic_over_algorithmic"))
for k in ("bf16_mode", "fp16_res32_mode", "vision_swin_tiny", "vision_swin_base", "vision_u8_pipeline", "id_tower", "bert_tiny", "fp32_parity_mode", "fp32x3_mode",
          "sustained", "cpu_baseline", "padded_token_layout", "with_item_dedup"):
    v = d.get(k)
    print(k, {kk: v[kk] for kk in v if kk in ("ms_per_step", "user_seq_per_s", "value", "error", "sample", "cores", "gemm_roofline_frac")} if isinstance(v, dict) else v)
print("eval", {k: (v.get("items_per_s") or v.get("users_per_s")) for k, v in d.get("eval", {}).items() if isinstance(v, dict)})
print(d["roofline"]["scoring"].get("pooled_8_ranks"))
PY
