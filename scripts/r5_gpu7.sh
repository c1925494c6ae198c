#!/bin/bash
# kernel-by-kernel timeline of one Swin-T step (order + durations), for the per-shape view of the weight-gradient launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
ROOT=$PWD
cd /tmp
MOREC_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --tower swin_tiny --batch 64 --steps 2 --warmup 1 > $ROOT/$O/kt.log 2>&1
echo "rc=$?"
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$f" > $ROOT/$O/timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]
# last step = the kernels after the last adamw burst but one: simply print the last third
n = len(rows)
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[n - n // 3 - 50:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} ms {d:9.1f} us  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?")):>8s} wg {r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")):>5s}  {short(r["Kernel_Name"])}')
PY
wc -l $ROOT/$O/timeline.txt
