"""Swin stage-1 MLP, forward fc1 + GELU and the backward down to dU: the tile kernels with a stored act' tensor against the streaming kernels
with the pre-activation recomputed (csrc/gemm_skinny_wide.hip): python scripts/mlp_recompute_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
L = _lib.lib()
dev, dt = "cuda", torch.bfloat16


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M, N, K, what in [(2207744, 384, 96, "Swin-T stage 1 (704 images)"), (1103872, 512, 128, "Swin-B stage 1 (352 images)"), (551936, 768, 192, "Swin-T stage 2 (704 images)")]:
    x = torch.randn(M, K, device=dev).to(dt); dy = torch.randn(M, K, device=dev).to(dt)
    w1 = (torch.randn(N, K, device=dev) * 0.1).to(dt); w2t = (torch.randn(N, K, device=dev) * 0.1).to(dt)
    b1 = torch.randn(N, device=dev); cs = torch.zeros(N, device=dev)
    g = torch.empty(M, N, device=dev, dtype=dt); pre = torch.empty(M, N, device=dev, dtype=dt); du = torch.empty(M, N, device=dev, dtype=dt)
    u = M * K * 2
    t_f_old = timed(lambda: ops.gemm_nt(x, w1, bias=b1, act=ops.ACT_GELU, out=g, aux_out=pre, aux_deriv=True))
    t_f_new = timed(lambda: ops.gemm_nt(x, w1, bias=b1, act=ops.ACT_GELU, out=g))
    L.morec_tuning_set(b"gemm_skinny", 1)
    t_f_tile = timed(lambda: ops.gemm_nt(x, w1, bias=b1, act=ops.ACT_GELU, out=g))
    L.morec_tuning_set(b"gemm_skinny", 0)
    t_b_old = timed(lambda: ops.gemm_nt(dy, w2t, dact=_lib.DACT_MUL, dact_in=pre, out=du, colsum_out=cs))
    t_b_new = timed(lambda: ops.mlp_dact_recompute(dy, w2t, x, w1, b1, colsum_out=cs))
    byt = lambda rd, wr: (rd + wr) * M * 2
    print(f"{what}: M = {M}, N = {N}, K = {K}")
    print(f"  fc1 + GELU: g and act' (tile kernel) {t_f_old:7.1f} us ({byt(K, 2 * N) / t_f_old / 1e6:5.2f} TB/s)   g only, tile kernel {t_f_tile:7.1f} us   "
          f"g only, streaming {t_f_new:7.1f} us ({byt(K, N) / t_f_new / 1e6:5.2f} TB/s)")
    print(f"  dU:  x act' read back (tile kernel) {t_b_old:7.1f} us ({byt(K + N, N) / t_b_old / 1e6:5.2f} TB/s)   pre-activation recomputed {t_b_new:7.1f} us "
          f"({byt(2 * K, N) / t_b_new / 1e6:5.2f} TB/s)")

