"""LayerNorm backward at the Swin-T stage shapes (704 images): python scripts/ln_bench_swin.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
dev, dt = "cuda", torch.bfloat16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for s in range(4):
    N, M = 96 * 2 ** s, 704 * 3136 // 4 ** s
    x = torch.randn(M, N, device=dev).to(dt); res = torch.randn(M, N, device=dev).to(dt)
    gm, bt = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    y, z, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-5, res=res)
    da = torch.randn(M, N, device=dev).to(dt); dr = torch.randn(M, N, device=dev).to(dt)
    dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    tf = timeit(lambda: ops.layernorm_fwd(x, gm, bt, 1e-5, res=res))
    tb = timeit(lambda: ops.layernorm_bwd(da, None, z, mean, rstd, gm, dg, db, dres=dr))
    row = M * N * 2 / 1e6
    print(f"C={N:4d} M={M:8d}: fwd {tf:7.1f} us {4 * row / tf:5.2f} TB/s | bwd {tb:7.1f} us {4 * row / tb:5.2f} TB/s (dy, z, dres -> dz)", flush=True)
