"""gemm8p tile height A/B at the bench shapes: launch time with 256-row tiles vs the automatic choice (and the forced heights)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import _lib, ops
from idvs.morec_amd._lib import ACT_GELU, DACT_MUL

L = _lib.lib()
dt = torch.float16
Ms = [int(x) for x in os.environ.get("TMR_M", "54919,51200,55808").split(",")]
for M in Ms:
    for N, K, kind in ((768, 768, "plain"), (2304, 768, "plain"), (3072, 768, "gelu"), (768, 3072, "plain"), (3072, 768, "dact"), (768, 2304, "plain")):
        a = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
        b = (torch.randn(N, K, device="cuda") * 0.5).to(dt)
        aux = torch.empty(M, N, device="cuda", dtype=dt) if kind == "gelu" else None
        din = torch.randn(M, N, device="cuda").to(dt) if kind == "dact" else None
        cs = torch.zeros(N, device="cuda") if kind == "dact" else None
        out = torch.empty(M, N, device="cuda", dtype=dt)
        row = []
        for mode in (0, 1, 224, 192, 0, 1):
            L.morec_tuning_set(b"gemm8p_tmr", mode)
            kw = dict(act=ACT_GELU, aux_out=aux, aux_deriv=True) if kind == "gelu" else (dict(dact=DACT_MUL, dact_in=din, colsum_out=cs) if kind == "dact" else {})
            for _ in range(3):
                ops.gemm_nt(a, b, out=out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(a, b, out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            row.append(f"{mode}:{us:.1f}us/{2.0 * M * N * K / us / 1e6:.0f}TF")
        print(f"M={M} N={N} K={K} {kind}: " + "  ".join(row), flush=True)
L.morec_tuning_set(b"gemm8p_tmr", 1)
