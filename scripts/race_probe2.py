"""Concurrency probe: Swin window-attention backward on the main stream beside a weight-gradient GEMM on a side stream (as in the step);
both outputs must be bit-stable from launch to launch.  python scripts/race_probe2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
from idvs.morec_amd.engine import _splitk
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
n_img, H, W, heads = 176, 56, 56, 3
C = heads * 32
M = n_img * H * W
side = torch.cuda.Stream()
for shift in (0, 3):
    desc = ops.swin_attn_desc(n_img, H, W, 7, shift, heads, 32, dt)
    qkv = torch.randn(M, 3 * C, device=dev).to(dt)
    table = torch.randn((2 * 7 - 1) ** 2, heads, device=dev) * 0.1
    bias_t = ops.swin_bias_expand(table, 7)
    dctx = torch.randn(M, C, device=dev).to(dt)
    ctx = ops.swin_attn_fwd(desc, qkv, bias_t)
    dy = torch.randn(M, 3 * C, device=dev).to(dt) * 0.1
    xn = torch.randn(M, C, device=dev).to(dt)
    gw_ref, dq_ref, cs_ref = None, None, None
    bad_tn = bad_at = bad_cs = 0
    for it in range(60):
        gw = torch.zeros(3 * C, C, device=dev)
        gb = torch.zeros(3 * C, device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.gemm_tn_(dy, xn, gw, split_m=_splitk(3 * C, C, M))
        dq = ops.swin_attn_bwd(desc, qkv, bias_t, ctx, dctx, torch.zeros_like(bias_t), dbqkv=gb)
        dx = ops.gemm_nt(dq, torch.randn(C, 3 * C, device=dev).to(dt))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if gw_ref is None:
            gw_ref, dq_ref, cs_ref = gw.clone(), dq.clone(), gb.clone()
        bad_tn += not torch.equal(gw, gw_ref)
        if not torch.equal(dq, dq_ref):
            bad_at += 1
            if bad_at <= 3:
                diff = (dq.float() != dq_ref.float()) | (torch.isnan(dq.float()) != torch.isnan(dq_ref.float()))
                idx = diff.nonzero()
                rows = idx[:, 0].unique()
                cols = idx[:, 1].unique()
                r0 = int(rows[0])
                img, rem = r0 // (H * W), r0 % (H * W)
                print(f"   it {it}: {idx.shape[0]} differing elements in {rows.numel()} rows (first rows {rows[:6].tolist()}: image {img}, y {rem // W}, x {rem % W}), columns {cols[:12].tolist()}... "
                      f"(n={cols.numel()}); got {dq[r0, cols[:4]].float().tolist()} ref {dq_ref[r0, cols[:4]].float().tolist()}", flush=True)
        bad_cs += not bool(torch.isfinite(gb).all())
    print(f"shift {shift}: TN GEMM mismatches {bad_tn}/60, attention bwd mismatches {bad_at}/60, non-finite bias sums {bad_cs}; finite: {bool(torch.isfinite(gw_ref).all())} {bool(torch.isfinite(dq_ref.float()).all())}", flush=True)
