"""Main-loop-dominated launches of the two eight-phase kernels: one round of 256 tiles, 128 K-tiles / token stages each, so that
prologue, epilogue and quantisation are < 5 % of the launch.   python scripts/mainloop_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
dev, dt = "cuda", torch.bfloat16
_lib.lib().morec_tuning_set(b"gemm8p", 2)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for L in (8192, 16384):
    a = torch.randn(4096, L, device=dev).to(dt); b = torch.randn(4096, L, device=dev).to(dt)
    out = torch.empty(4096, 4096, device=dev, dtype=dt)
    us = timeit(lambda: ops.gemm_nt(a, b, out=out))
    fl = 2.0 * 4096 * 4096 * L
    print(f"NT  C[4096,4096] = A[4096,{L}] B[4096,{L}]^T : {us:7.1f} us {fl / us / 1e6:7.1f} TF/s   ({us * 1e3 / (L // 64):6.1f} ns per K-tile)")
    dy = torch.randn(L, 4096, device=dev).to(dt); x = torch.randn(L, 4096, device=dev).to(dt)
    o32 = torch.zeros(4096, 4096, device=dev)
    us = timeit(lambda: ops.gemm_tn_(dy, x, o32, split_m=1, accumulate=False))
    print(f"TN  C[4096,4096] = DY[{L},4096]^T X[{L},4096]  : {us:7.1f} us {fl / us / 1e6:7.1f} TF/s   ({us * 1e3 / (L // 64):6.1f} ns per stage)")
    yt = torch.matmul(a, b.t()); us = timeit(lambda: torch.matmul(a, b.t()))
    print(f"lib torch.matmul(A, B^T)                          : {us:7.1f} us {fl / us / 1e6:7.1f} TF/s")
    us = timeit(lambda: torch.matmul(dy.t(), x))
    print(f"lib torch.matmul(DY^T, X)                         : {us:7.1f} us {fl / us / 1e6:7.1f} TF/s")
