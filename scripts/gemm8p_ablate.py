"""Where a 256 x 256 eight-phase GEMM tile spends its time: the same launch with (1) global stores skipped, (2) the whole epilogue
skipped, (4) the main loop cut to two K-tiles -- tuning key "gemm8p_debug".   python scripts/gemm8p_ablate.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
from idvs.morec_amd._lib import ACT_GELU
dev, dt = "cuda", torch.bfloat16
L = _lib.lib()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
L.morec_tuning_set(b"gemm8p", 2)
for name, N, K, kind in [("qkv", 2304, 768, "bias"), ("o", 768, 768, "plain"), ("fc1+gelu", 3072, 768, "gelu"), ("fc2", 768, 3072, "plain"), ("d_fc2(dact+cs)", 3072, 768, "dact")]:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    kw = {}
    if kind == "bias": kw = dict(bias=torch.zeros(N, device=dev))
    if kind == "gelu": kw = dict(bias=torch.zeros(N, device=dev), act=ACT_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt))
    if kind == "dact": kw = dict(dact=ACT_GELU, dact_in=torch.randn(M, N, device=dev).to(dt), colsum_out=torch.zeros(N, device=dev))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    row = []
    for dbg in (0, 1, 2, 4, 5, 6):
        L.morec_tuning_set(b"gemm8p_debug", dbg)
        row.append(min(timeit(lambda: ops.gemm_nt(a, b, out=out, **kw)) for _ in range(3)))
    L.morec_tuning_set(b"gemm8p_debug", 0)
    print(f"{name:15s} N={N:5d} K={K:5d} tiles {tiles:5d} ({tiles/256:.2f} rounds): full {row[0]:7.1f} | no stores {row[1]:7.1f} | no epilogue {row[2]:7.1f} | "
          f"2 K-tiles: full {row[3]:7.1f} no stores {row[4]:7.1f} no epilogue {row[5]:7.1f}  (us)", flush=True)
