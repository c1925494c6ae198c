"""Time every GEMM shape of one Swin-T training step (B = 64 users x 11 images) in isolation: python scripts/swin_gemm_shapes.py [n_img [C0]]
(the MLP pair is timed in the stored-act' form; the recompute form of stage 1: scripts/mlp_recompute_bench.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
from idvs.morec_amd._lib import ACT_GELU
from idvs.morec_amd.engine import _splitk
dev, dt = "cuda", torch.bfloat16
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 704
C0 = int(sys.argv[2]) if len(sys.argv) > 2 else 96          # 128: Swin-B (depths 2 / 2 / 18 / 2)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0.0
for s in range(4):
    C, M = C0 * 2 ** s, n_img * 3136 // 4 ** s
    reps = ([2, 2, 6, 2] if C0 == 96 else [2, 2, 18, 2])[s]
    for name, N, K, kind in [("qkv", 3 * C, C, "nt"), ("o", C, C, "nt"), ("fc1+gelu", 4 * C, C, "gelu"), ("fc2", C, 4 * C, "nt"),
                             ("d_fc2(dact)", 4 * C, C, "dact"), ("d_fc1", C, 4 * C, "nt"), ("d_o", C, C, "nt"), ("d_qkv", C, 3 * C, "nt"),
                             ("w_qkv", 3 * C, C, "tn"), ("w_o", C, C, "tn"), ("w_fc1", 4 * C, C, "tn"), ("w_fc2", C, 4 * C, "tn")]:
        if kind == "tn":
            dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
            out = torch.zeros(N, K, device=dev)
            sp = _splitk(N, K, M)
            us = timeit(lambda: ops.gemm_tn_(dy, x, out, split_m=sp))
            byt = (M * N + M * K) * 2
        else:
            a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
            out = torch.empty(M, N, device=dev, dtype=dt)
            kw = {}
            byt = (M * K + M * N) * 2
            if kind == "gelu":
                kw = dict(bias=torch.zeros(N, device=dev), act=ACT_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt)); byt += M * N * 2
            if kind == "dact":
                kw = dict(dact=ACT_GELU, dact_in=torch.randn(M, N, device=dev).to(dt)); byt += M * N * 2
            us = timeit(lambda: ops.gemm_nt(a, b, out=out, **kw))
        fl = 2.0 * M * N * K
        tot += us * reps
        print(f"s{s} {name:12s} M={M:8d} N={N:5d} K={K:5d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {byt / us / 1e3:7.1f} GB/s (alg)  x{reps}")
print(f"total GEMM time per step (these shapes): {tot / 1e3:.2f} ms")
