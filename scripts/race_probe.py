"""Repeat-launch determinism probe: a kernel without atomics must give bit-identical outputs on every launch; a mismatch = a race.
python scripts/race_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
def probe(name, fn, n=150):
    ref = [t.clone() for t in fn()]
    bad = 0
    for i in range(n):
        out = fn()
        if any(not torch.equal(a, b) for a, b in zip(out, ref)):
            bad += 1
    nan = any(bool(torch.isnan(t.float()).any()) for t in ref)
    print(f"{name:60s} mismatching launches {bad}/{n}  nan={nan}", flush=True)
for M, N, K, bias in [(551936, 96, 48, True), (551936, 288, 96, True), (551936, 96, 96, False), (551936, 96, 384, False), (551936, 96, 288, False),
                      (2207744, 288, 96, True), (2207744, 96, 384, False), (100003, 128, 128, False)]:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
    bv = torch.randn(N, device=dev) if bias else None
    # other work in flight on a second stream, as in the step (weight-gradient GEMMs beside the dX chain)
    side = torch.cuda.Stream()
    x = torch.randn(8192, 1024, device=dev).to(dt); y = torch.randn(8192, 1024, device=dev).to(dt); o2 = torch.zeros(1024, 1024, device=dev)
    def fn():
        with torch.cuda.stream(side):
            ops.gemm_tn_(x, y, o2, split_m=4)
        return (ops.gemm_nt(a, b, bias=bv),)
    probe(f"gemm_skinny {M} x {N} x {K} bias={bias}", fn, 60)
    torch.cuda.synchronize()
# Swin window attention
n_img, H, W, heads = 176, 56, 56, 3
C = heads * 32
for shift in (0, 3):
    desc = ops.swin_attn_desc(n_img, H, W, 7, shift, heads, 32, dt)
    qkv = torch.randn(n_img * H * W, 3 * C, device=dev).to(dt)
    table = torch.randn((2 * 7 - 1) ** 2, heads, device=dev) * 0.1
    bias_t = ops.swin_bias_expand(table, 7)
    dctx = torch.randn(n_img * H * W, C, device=dev).to(dt)
    probe(f"swin_attn_fwd shift={shift}", lambda: (ops.swin_attn_fwd(desc, qkv, bias_t),), 40)
    ctx = ops.swin_attn_fwd(desc, qkv, bias_t)
    probe(f"swin_attn_bwd shift={shift}", lambda: (ops.swin_attn_bwd(desc, qkv, bias_t, ctx, dctx),), 40)
