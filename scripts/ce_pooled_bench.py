"""The scoring kernels (fused in-batch debiased CE forward + backward) at the one-GPU size (Nc = 2 688 columns) and at the 8-GPU
pooled size (Nc = 21 504, emulated on one GPU: this rank's 2 560 rows against eight ranks' worth of item vectors, col_offset =
3 * 2 688), bf16, D = 512: time, executed TFLOP/s and algorithmic GB/s (SURVEY.md §8d byte counts).  Run it under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` for the counter view.   python scripts/ce_pooled_bench.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from idvs.morec_amd import ops, _lib
dev, dt = "cuda", torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # tuning key "ce8p": 0 automatic, 1 the 128 x 128 kernels, 2 the 256 x 256 eight-phase kernels
_lib.lib().morec_tuning_set(b"ce8p", MODE)
B, S, D = 128, 20, int(sys.argv[3]) if len(sys.argv) > 3 else 512
RANKS = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 4, 8]      # pooled sizes to run (a counter pass wants ONE)
print(f"# ce8p mode {MODE}, D = {D}")
for ranks, rank in [(r, min(3, r - 1)) for r in RANKS]:
    Nr, Nc = B * S, ranks * B * (S + 1)
    g = torch.Generator(device=dev).manual_seed(ranks)
    P = (torch.randn(Nr, D, device=dev, generator=g) * 0.3).to(dt); E = (torch.randn(Nc, D, device=dev, generator=g) * 0.3).to(dt)
    ids = torch.randint(1, 80000, (Nc,), device=dev, generator=g, dtype=torch.int32)
    row_ids = ids[rank * B * (S + 1):(rank + 1) * B * (S + 1)].contiguous()
    logpop = torch.randn(Nc, device=dev, generator=g) - 9.0
    col_valid = torch.ones(Nc, device=dev, dtype=torch.uint8); row_valid = torch.ones(Nr, device=dev, dtype=torch.uint8)
    desc = ops.ce_desc(B, S, D, Nc, rank * B * (S + 1), dt, dE_fp32=(ranks > 1))
    ws = ops.ce_workspace(desc, dev)
    def fwd(): return ops.inbatch_ce_fwd(desc, P, E, row_ids, ids, logpop, col_valid, row_valid, ws)
    loss_sum, lse, _ = fwd()
    bdesc = ops.ce_desc(B, S, D, Nc, rank * B * (S + 1), dt, dE_fp32=(ranks > 1), ws_from_fwd=True)     # as engine.ce_backward: the forward's tables reused
    def bwd(): return ops.inbatch_ce_bwd(bdesc, P, E, row_ids, ids, logpop, col_valid, row_valid, lse, None, 1.0 / Nr, ws)
    bwd(); torch.cuda.synchronize()
    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters * 1e3
    fl_f, fl_b = 2.0 * Nr * Nc * D, 3 * 2.0 * Nr * Nc * D            # bwd: recompute + dP + dE
    by_f = (Nr + Nc) * D * 2 + 13 * Nc + 8 * Nr                        # reads P, E + bookkeeping, writes loss / lse
    by_b = (Nr + Nc) * D * 2 + 4 * Nr + Nr * D * 2 + Nc * D * (4 if ranks > 1 else 2)
    print(f"Nc = {Nc:6d} ({ranks} rank(s)): fwd {res['fwd']:7.1f} us = {fl_f / res['fwd'] / 1e6:6.1f} TFLOP/s, {by_f / res['fwd'] / 1e3:7.1f} GB/s algorithmic | "
          f"bwd {res['bwd']:7.1f} us = {fl_b / res['bwd'] / 1e6:6.1f} TFLOP/s, {by_b / res['bwd'] / 1e3:7.1f} GB/s algorithmic | "
          f"arithmetic intensity fwd {fl_f / by_f:6.0f} FLOP/B (machine balance 2.5e15 / 8e12 = 312): MFMA-bound, not HBM-bound", flush=True)
