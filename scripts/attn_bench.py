"""BERT-tower attention (bf16 MFMA kernels) on the unpadded token layout: time per launch and effective HBM rate.
usage: python scripts/attn_bench.py [n_seq]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import ops  # noqa: E402


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
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 2688
    heads, dh = 12, 64
    H = heads * dh
    rng = np.random.default_rng(0)
    for label, lens in (("T=30 fixed", np.full(n_seq, 30)), ("ragged 8..30", rng.integers(8, 31, n_seq))):
        cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
        M = int(lens.sum())
        qkv = torch.randn(M, 3 * H, device="cuda").to(torch.bfloat16)
        dctx = torch.randn(M, H, device="cuda").to(torch.bfloat16)
        keep = torch.ones(M, device="cuda")
        d = ops.attn_desc(n_seq, 30, heads, dh, False, dh ** -0.5, -10000.0, torch.bfloat16, 0.1, 77, cu_seqlens=cu)
        tf = timeit(lambda: ops.attn_fwd(d, qkv, keep))
        tb = timeit(lambda: ops.attn_bwd(d, qkv, keep, dctx))
        dbias = torch.zeros(3 * H, device="cuda")
        tbb = timeit(lambda: ops.attn_bwd(d, qkv, keep, dctx, dbias=dbias))
        bf, bb = M * H * 2 * 4, M * H * 2 * 7
        print(f"{label}: M={M}  fwd {tf:.1f} us ({bf / tf / 1e3:.0f} GB/s)  bwd {tb:.1f} us ({bb / tb / 1e3:.0f} GB/s)  "
              f"bwd + q|k|v bias gradient {tbb:.1f} us (incl. the fold of the per-sequence sums)")


if __name__ == "__main__":
    main()
