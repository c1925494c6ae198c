"""d(b1) fused into the dGELU GEMM epilogue vs the separate column-sum pass: correctness + time per launch.
usage: python scripts/colsum_fuse_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import ops  # noqa: E402

ACT_GELU = 1


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 29312
    dev = "cuda"
    torch.manual_seed(0)
    for dtype in (torch.bfloat16, torch.float32):
        for (N, K) in ((3072, 768), (384, 96), (768, 192), (1536, 384)):
            dy = torch.randn(M, K, device=dev).to(dtype)
            w = (torch.randn(N, K, device=dev) * 0.05).to(dtype)
            u = torch.randn(M, N, device=dev).to(dtype)
            b0 = torch.zeros(N, device=dev)
            b1 = torch.zeros(N, device=dev)
            du0 = ops.gemm_nt(dy, w, dact=ACT_GELU, dact_in=u)
            ops.colsum_(du0, b0)
            du1 = ops.gemm_nt(dy, w, dact=ACT_GELU, dact_in=u, colsum_out=b1)
            ref = du0.float().sum(0)
            same = torch.equal(du0, du1)
            e_old = float((b0 - ref).abs().max() / ref.abs().max())
            e_new = float((b1 - ref).abs().max() / ref.abs().max())
            t_g = timeit(lambda: ops.gemm_nt(dy, w, dact=ACT_GELU, dact_in=u))
            t_s = timeit(lambda: (ops.gemm_nt(dy, w, dact=ACT_GELU, dact_in=u), ops.colsum_(du0, b0)))
            t_f = timeit(lambda: ops.gemm_nt(dy, w, dact=ACT_GELU, dact_in=u, colsum_out=b1))
            print(f"{str(dtype)[6:]:9s} M={M} N={N} K={K}: C identical {same}; colsum rel err separate {e_old:.2e} fused {e_new:.2e}; "
                  f"gemm {t_g:.1f} us, gemm + colsum {t_s:.1f} us, fused {t_f:.1f} us")


if __name__ == "__main__":
    main()
