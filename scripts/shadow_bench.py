"""Plain gemm8p launches at the bench shapes: used with MOREC_HIP_LIB pointing at experiment builds (dummy VALU work in the read / DMA
segments of the main loop, -DG8_EXP_SHADOW=n) to see what VALU work in the partner wave's MFMA shadow costs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import _lib, ops
from idvs.morec_amd._lib import ACT_GELU

dt = torch.float16
M = int(os.environ.get("SB_M", "54919"))
tag = os.path.basename(_lib.LIB_PATH)
for N, K, kind in ((3072, 768, "plain"), (3072, 768, "gelu"), (768, 3072, "plain"), (2304, 768, "plain"), (768, 768, "plain")):
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    b = (torch.randn(N, K, device="cuda") * 0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    aux = torch.empty(M, N, device="cuda", dtype=dt) if kind == "gelu" else None
    kw = dict(act=ACT_GELU, aux_out=aux, aux_deriv=True) if kind == "gelu" else {}
    res = []
    for rep in range(3):
        for _ in range(3):
            ops.gemm_nt(a, b, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_nt(a, b, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    us = min(res)
    print(f"{tag} M={M} N={N} K={K} {kind}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s  (runs {' '.join(f'{r:.1f}' for r in res)})", flush=True)
