"""gemm8p tail split against the unsplit kernel on ALTERNATING inputs (a stale partial from the previous launch would be a large
error, not a rounding difference), optionally with a big unrelated kernel in between.   python scripts/tail_split_repeat.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
L = _lib.lib()
L.morec_tuning_set(b"gemm8p_tail_split", 1)
L.morec_tuning_set(b"gemm8p", 2)
dev, dt = "cuda", torch.bfloat16
for (M, N, K) in [(51200, 768, 3072), (50937, 768, 3072), (51200, 768, 2304)]:
    sets = []
    for i in range(3):
        a = (torch.randn(M, K, device=dev) * (1 + i)).to(dt); b = torch.randn(N, K, device=dev).to(dt)
        L.morec_tuning_set(b"gemm8p_debug", 128)       # no tail split
        ref = ops.gemm_nt(a, b).clone()
        sets.append((a, b, ref))
    L.morec_tuning_set(b"gemm8p_debug", 0)
    filler = torch.randn(64 * 1024 * 1024, device=dev)
    for mode in ("back to back", "with a 256 MB elementwise kernel in between"):
        worst, bad = 0.0, 0
        for it in range(30):
            a, b, ref = sets[it % 3]
            o = ops.gemm_nt(a, b)
            if mode != "back to back":
                filler.mul_(1.0001)
            d = (o.float() - ref.float()).abs()
            rel = float(d.max().item()) / float(ref.float().abs().max().item())
            worst = max(worst, rel)
            bad += int((d > 0.02 * ref.float().abs().max()).sum().item())
        print(f"M={M} N={N} K={K} {mode}: worst |split - unsplit| / max|ref| = {worst:.2e}, elements off by > 2 % of the range: {bad}", flush=True)
