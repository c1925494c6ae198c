"""Write-bound GEMM (Swin stage-1 FFN-up: M = 2.2 M, N = 384, K = 96, GELU + second output) under the four output-store cache
policies of gemm8p (debug bits 4-5: 0 default, 1 nt, 2 sc1, 3 sc1 nt).   python scripts/store_policy_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
from idvs.morec_amd._lib import ACT_GELU
dev, dt = "cuda", torch.bfloat16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, kind) in [(2207744, 384, 96, "gelu"), (551936, 768, 192, "gelu"), (51200, 3072, 768, "gelu"), (51200, 2304, 768, "bias")]:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt); aux = torch.empty(M, N, device=dev, dtype=dt)
    kw = dict(bias=torch.zeros(N, device=dev))
    if kind == "gelu":
        kw.update(act=ACT_GELU, aux_out=aux, aux_deriv=True)
    for cp, name in enumerate(("default", "nt", "sc1", "sc1 nt")):
        _lib.lib().morec_tuning_set(b"gemm8p_debug", cp << 4)
        _lib.lib().morec_tuning_set(b"gemm8p", 2)
        us = timeit(lambda: ops.gemm_nt(a, b, out=out, **kw))
        wr = M * N * 2 * (2 if kind == "gelu" else 1)
        print(f"M={M:8d} N={N:5d} K={K:4d} {kind:5s} stores {name:7s}: {us:8.1f} us   {wr / us / 1e6:5.2f} TB/s written", flush=True)
_lib.lib().morec_tuning_set(b"gemm8p_debug", 0)
