"""Weight-gradient GEMM (dW = dY^T X) on Swin-T / Swin-B shapes whose outputs are narrower than the 256 x 256 tile of gemm_tn8p: run once with
MOREC_TN8P_SKIP=0 (every block of the tile multiplied) and once without (blocks outside the matrix skipped): python scripts/tn_skip_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, engine
dev, dt = "cuda", torch.bfloat16
shapes = [(2207744, 96, 384, "T stage 1 dW(fc2)"), (2207744, 384, 96, "T stage 1 dW(fc1)"), (2207744, 288, 96, "T stage 1 dW(qkv)"), (2207744, 96, 96, "T stage 1 dW(o)"),
          (551936, 192, 768, "T stage 2 dW(fc2)"), (551936, 768, 192, "T stage 2 dW(fc1)"), (551936, 576, 192, "T stage 2 dW(qkv)"), (551936, 192, 192, "T stage 2 dW(o)"),
          (137984, 384, 1536, "T stage 3 dW(fc2)"), (137984, 1152, 384, "T stage 3 dW(qkv)"), (1103872, 128, 512, "B stage 1 dW(fc2)"), (1103872, 384, 128, "B stage 1 dW(qkv)")]
print("MOREC_TN8P_SKIP =", os.environ.get("MOREC_TN8P_SKIP", "(default: on)"))
for M, N, K, what in shapes:
    dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt); out = torch.zeros(N, K, device=dev)
    split = engine._splitk(N, K, M)
    for _ in range(3): ops.gemm_tn_(dy, x, out, split_m=split)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm_tn_(dy, x, out, split_m=split)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{what:20s} [{N:4d} x {K:4d}] over {M:8d} rows, split {split:3d}: {us:7.1f} us  ({(M * (N + K)) * 2 / us / 1e6:5.2f} TB/s, {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s)")
    del dy, x
