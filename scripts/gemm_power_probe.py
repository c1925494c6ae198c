"""Is a GEMM launch power-limited?  The same kernel, same shape, same instruction stream on operands of different bit activity: N(0, 0.5) random,
small integers, a constant, zeros.  A kernel bound by its schedule runs them in the same time; a part that holds a power cap runs the quiet data at a
higher clock (MI355X_MICROARCH.md, "DVFS give-back").   python scripts/gemm_power_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import _lib, ops  # noqa: E402

dev, dt = "cuda", torch.float16
M = 54919


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for (N, K) in ((3072, 768), (768, 3072)):
    for kern, mode in (("gemm8p", 1), ("gemm2w", 2)):
        _lib.check(_lib.lib().morec_tuning_set(b"gemm2w", mode), "tuning")
        row = []
        for fill in ("randn", "int", "const", "zero"):
            if fill == "randn":
                a, b = (torch.randn(M, K, device=dev) * 0.5).to(dt), (torch.randn(N, K, device=dev) * 0.5).to(dt)
            elif fill == "int":
                a, b = torch.randint(-3, 4, (M, K), device=dev).to(dt), torch.randint(-3, 4, (N, K), device=dev).to(dt)
            elif fill == "const":
                a, b = torch.full((M, K), 0.5, device=dev, dtype=dt), torch.full((N, K), 0.25, device=dev, dtype=dt)
            else:
                a, b = torch.zeros(M, K, device=dev, dtype=dt), torch.zeros(N, K, device=dev, dtype=dt)
            out = torch.empty(M, N, device=dev, dtype=dt)
            us = timeit(lambda: ops.gemm_nt(a, b, out=out))
            row.append(f"{fill} {us:6.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF")
        print(f"{kern} M={M} N={N} K={K}: " + " | ".join(row), flush=True)
_lib.check(_lib.lib().morec_tuning_set(b"gemm2w", 0), "tuning")
