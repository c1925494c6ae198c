"""fp32 attention (exact-fp32 parity mode / fp32x3 mode) at the BERT-base and SASRec shapes of the bench step: time per launch and effective HBM
rate.  Run once with MOREC_ATTN_F32_MFMA=0 (VALU kernels of attention.hip) and once without (attention_f32mfma.hip)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import ops  # noqa: E402
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rng = np.random.default_rng(0)
for label, n_seq, T, heads, dh, causal, lens in (("BERT-base ragged 8..30", 2688, 30, 12, 64, False, rng.integers(8, 31, 2688)),
                                                 ("SASRec S=20 D=512", 128, 20, 2, 256, True, None)):
    H = heads * dh
    if lens is not None:
        cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda(); M = int(lens.sum())
    else:
        cu, M = None, n_seq * T
    qkv = torch.randn(M, 3 * H, device="cuda"); dctx = torch.randn(M, H, device="cuda")
    keep = torch.ones(M, device="cuda")
    d = ops.attn_desc(n_seq, T, heads, dh, causal, dh ** -0.5, -1e9, torch.float32, 0.1, 77, cu_seqlens=cu, total_rows=M if cu is not None else 0)
    tf = timeit(lambda: ops.attn_fwd(d, qkv, keep)); tb = timeit(lambda: ops.attn_bwd(d, qkv, keep, dctx))
    print(f"{label}: {M} rows  fwd {tf:7.1f} us ({M * 4 * H * 4 / tf / 1e6:.2f} TB/s)  bwd {tb:7.1f} us ({M * 8 * H * 4 / tb / 1e6:.2f} TB/s)   MOREC_ATTN_F32_MFMA={os.environ.get('MOREC_ATTN_F32_MFMA', '1')}")
