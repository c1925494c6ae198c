"""Attention at T = 50 (abstracts / bodies): attention_mfma64.hip vs the VALU fallback (MOREC_ATTN_MFMA64=0 in a second process), BERT-base heads."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import ops

n_seq, T, nh, dh = 2688, 50, 12, 64
H = nh * dh
dt = torch.float16
qkv = (0.5 * torch.randn(n_seq * T, 3 * H, device="cuda")).to(dt)
dctx = (0.5 * torch.randn(n_seq * T, H, device="cuda")).to(dt)
keep = torch.ones(n_seq, T, device="cuda")
desc = ops.attn_desc(n_seq, T, nh, dh, False, 1 / math.sqrt(dh), ops.FLT_MIN_MASK, dt, 0.1, 5)
for name, fn in (("fwd", lambda: ops.attn_fwd(desc, qkv, keep)), ("bwd", lambda: ops.attn_bwd(desc, qkv, keep, dctx))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    byt = n_seq * T * H * 2 * (4 if name == "fwd" else 8)
    print(f"MOREC_ATTN_MFMA64={os.environ.get('MOREC_ATTN_MFMA64', '1')} {name}: {us:.1f} us, {byt / us / 1e6:.2f} TB/s algorithmic")
