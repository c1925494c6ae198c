"""Run-to-run scatter of the small vision / text fp32 training runs of tests/test_deterministic_gpu.py in the DEFAULT (atomic) mode: losses of n runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_deterministic_gpu as t
tower, dtype, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
det = len(sys.argv) > 4 and sys.argv[4] == "det"
ref = None
for i in range(n):
    l, st = t._run(tower, dtype, 5, det)
    if ref is None: ref = (l, st)
    dp = max(float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)) for x, y in zip(st[::3], ref[1][::3]))
    print(f"run {i}: losses {' '.join(f'{v:.7f}' for v in l)}  max|dl| vs run 0 {max(abs(u - v) for u, v in zip(l, ref[0])):.2e}  param dist {dp:.2e}", flush=True)
