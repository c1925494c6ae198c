"""Swin window attention forward / backward at the four Swin-T stage shapes of the bench (704 images): launch time and the
algorithmic HBM rate (fwd: qkv in, ctx out; bwd: qkv + dctx in, dqkv out).   python scripts/swin_attn_bench.py [n_img]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 704
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
    C, HW, heads = 96 * 2 ** s, 56 // 2 ** s, 3 * 2 ** s
    M = n_img * HW * HW
    for shift in (0, 3):
        if HW == 7 and shift:
            continue
        qkv = (torch.randn(M, 3 * C, device=dev) * 0.5).to(dt)
        bias_t = torch.randn(heads, 49, 49, device=dev) * 0.1
        desc = ops.swin_attn_desc(n_img, HW, HW, 7, shift, heads, 32, dt)
        tf = timeit(lambda: ops.swin_attn_fwd(desc, qkv, bias_t))
        ctx = ops.swin_attn_fwd(desc, qkv, bias_t)
        dctx = torch.randn_like(ctx)
        dbias = None if os.environ.get("SAB_NODBIAS") else torch.zeros_like(bias_t)      # SAB_NODBIAS=1: what the dbias tail of the backward costs
        tb = timeit(lambda: ops.swin_attn_bwd(desc, qkv, bias_t, ctx, dctx, dbias))
        bf, bb = M * 4 * C * 2, M * 7 * C * 2
        print(f"stage {s} C={C:4d} M={M:8d} shift={shift}: fwd {tf:7.1f} us {bf / tf / 1e6:5.2f} TB/s | bwd {tb:7.1f} us {bb / tb / 1e6:5.2f} TB/s", flush=True)
