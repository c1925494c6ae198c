"""Streaming GEMM for narrow outputs vs the tile kernels on the Swin-T stage-1 / stage-2 shapes (704 images): python scripts/skinny_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
L = _lib.lib()
dev, dt = "cuda", torch.bfloat16
shapes = [(2207744, 288, 96, "stage 1 q|k|v projection (N = 288)"), (2207744, 96, 48, "patch embed"), (2207744, 96, 96, "o_proj / its dX"), (2207744, 96, 384, "fc2 / dX of fc1"), (2207744, 96, 288, "dX of qkv"),
          (551936, 192, 192, "stage 2 o_proj (N = 192: tile kernels)"), (1126400, 128, 128, "Swin-B stage 1 o_proj (352 images)"), (1126400, 128, 384, "Swin-B dX of qkv")]
for M, N, K, what in shapes:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt); out = torch.empty(M, N, device=dev, dtype=dt)
    res = []
    for mode in (1, 0):
        L.morec_tuning_set(b"gemm_skinny", mode)
        for _ in range(3): ops.gemm_nt(a, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_nt(a, b, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        byt = (M * K + N * K + M * N) * 2
        res.append((us, byt / us / 1e6))
    print(f"{M:8d} x {N:3d} x {K:3d}  {what:38s} tile kernels {res[0][0]:7.1f} us ({res[0][1]:5.2f} TB/s)   streaming {res[1][0]:7.1f} us ({res[1][1]:5.2f} TB/s)")
L.morec_tuning_set(b"gemm_skinny", 0)
