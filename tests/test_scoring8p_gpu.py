"""-m gpu: the scoring kernels on the eight-phase 256 x 256 main loop (``csrc/inbatch_ce8p.hip``: ``ce8p_kernel<fwd / bwd>`` + prep,
dE as one NT GEMM of dlogit^T, dP through the transposing GEMM) against the CPU oracle (``oracle/morec_oracle/nn_ref.py::
inbatch_ce_loss`` = ``T/model/model.py:32-33,45-67``, fp64 on the same bf16-rounded inputs) and against the 128 x 128 kernels they
replace at the pooled-negative sizes.  Forced with ``morec_tuning_set("ce8p", 2)`` so that the small oracle-sized cases run on them too
(automatic selection needs >= 96 tiles of 256 x 256).  Tolerances (bf16 operands): loss 2e-2 relative, dP / dE 3e-2 of the max-abs --
the bounds of ``test_kernels_gpu.py::test_inbatch_ce``."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from idvs.morec_amd import _lib, ops  # noqa: E402
from test_kernels_gpu import _ce_case, rel, rnd  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def _mode(v):
    assert _lib.lib().morec_tuning_set(b"ce8p", v) == 0


def _inputs(B, S, D, item_num, n_ranks, rank, seed=None):
    from morec_oracle import bookkeeping as bk
    ids_all, lm_all, pop = _ce_case(B, S, D, item_num, seed=(B + S) if seed is None else seed, n_ranks=n_ranks)
    ids, lm = ids_all[rank * B:(rank + 1) * B], lm_all[rank * B:(rank + 1) * B]
    t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(d)      # noqa: E731
    args = (t(ids.reshape(-1), torch.int32), t(ids_all.reshape(-1), torch.int32), t(bk.log_pop(pop, ids_all), torch.float32),
            t(bk.column_valid(lm_all), torch.uint8), t(lm.reshape(-1) != 0, torch.uint8))
    return ids_all, lm_all, pop, ids, lm, args


def _run(desc, P, E, args, n_valid, dE_fp32=False):
    ws = ops.ce_workspace(desc, DEV)
    loss_sum, lse, row_loss = ops.inbatch_ce_fwd(desc, P, E, *args, ws)
    # backward on the forward's untouched workspace (ws_from_fwd: the flag table / positive logits are reused) ...
    reuse = ops.CeDesc(desc.B, desc.S, desc.D, desc.Nc, desc.col_offset, desc.dtype, desc.dE_fp32, 1)
    dP_r, dE_r = ops.inbatch_ce_bwd(reuse, P, E, *args, lse, None, 1.0 / n_valid, ws)
    # ... and on a workspace that is plain scratch (everything rebuilt): the same numbers, bit for bit
    ws.fill_(0xA5)
    dP, dE = ops.inbatch_ce_bwd(desc, P, E, *args, lse, None, 1.0 / n_valid, ws)
    assert torch.equal(dP, dP_r) and torch.equal(dE, dE_r)
    return loss_sum.item() / n_valid, lse, row_loss, dP, dE


@pytest.mark.parametrize("cfg", [(16, 20, 512, 200, 1, 0), (128, 20, 512, 5000, 1, 0), (8, 20, 128, 100, 4, 2), (8, 10, 2048, 60, 2, 1),
                                 (24, 7, 192, 40, 3, 0), (40, 20, 512, 300, 2, 1)])
def test_scoring8p_vs_oracle(cfg):
    import morec_oracle as orc
    B, S, D, item_num, n_ranks, rank = cfg
    ids_all, lm_all, pop, ids, lm, args = _inputs(*cfg)
    Nc, off = ids_all.size, rank * B * (S + 1)
    assert (B * S) % 8 == 0 and Nc % 8 == 0
    E, P = rnd(Nc, D, dt=BF, scale=0.5), rnd(B * S, D, dt=BF, scale=0.5, seed=1)
    n_valid = int((lm_all != 0).sum())
    Pc, Ec = P.double().cpu().requires_grad_(True), E.double().cpu().requires_grad_(True)
    loss_ref = orc.inbatch_ce_loss(Pc, Ec, ids, lm, pop, S, pool_ids=ids_all, pool_log_mask=lm_all, col_offset=off, n_valid_total=n_valid)
    loss_ref.backward()
    try:
        _mode(2)
        loss, lse, row_loss, dP, dE = _run(ops.ce_desc(B, S, D, Nc, off, BF), P, E, args, n_valid)
        _, _, _, dP2, dE32 = _run(ops.ce_desc(B, S, D, Nc, off, BF, dE_fp32=True), P, E, args, n_valid)
        _mode(1)
        loss_o, lse_o, row_loss_o, dP_o, dE_o = _run(ops.ce_desc(B, S, D, Nc, off, BF), P, E, args, n_valid)
    finally:
        _mode(0)
    assert abs(loss - loss_ref.item()) < 2e-2 * max(1.0, abs(loss_ref.item()))
    assert rel(dP.cpu(), Pc.grad) < 3e-2 and rel(dE.cpu(), Ec.grad) < 3e-2
    assert dE32.dtype == torch.float32 and rel(dE32.cpu(), Ec.grad) < 3e-2 and torch.equal(dP2, dP)
    # same logits as the 128 x 128 kernels up to the fp32 summation order of the dot products
    valid = args[4].bool()
    assert float((lse - lse_o).abs()[valid].max()) < 2e-3 and float((row_loss - row_loss_o).abs().max()) < 2e-3
    assert abs(loss - loss_o) < 1e-4 * max(1.0, abs(loss_o))
    assert rel(dP, dP_o.double()) < 2e-2 and rel(dE, dE_o.double()) < 2e-2
    # rows of padded positions get no gradient at all
    assert float(dP.float()[~valid].abs().max() if (~valid).any() else 0.0) == 0.0


def test_scoring8p_pooled_bench_size_matches_the_128_kernels():
    """The 8-rank pooled size of the benchmark (Nr = 2560 rows against Nc = 21 504 columns, D = 512), automatic selection: the new
    kernels against the ones they replace on identical inputs (the CPU oracle needs minutes at this size), and run-to-run determinism."""
    cfg = (128, 20, 512, 60000, 8, 3)
    B, S, D = cfg[:3]
    ids_all, lm_all, pop, ids, lm, args = _inputs(*cfg)
    Nc, off = ids_all.size, cfg[5] * B * (S + 1)
    E, P = rnd(Nc, D, dt=BF, scale=0.3), rnd(B * S, D, dt=BF, scale=0.3, seed=1)
    n_valid = int((lm_all != 0).sum())
    desc = ops.ce_desc(B, S, D, Nc, off, BF, dE_fp32=True)
    try:
        _mode(0)
        a = _run(desc, P, E, args, n_valid)
        b = _run(desc, P, E, args, n_valid)
        _mode(1)
        o = _run(desc, P, E, args, n_valid)
    finally:
        _mode(0)
    assert abs(a[0] - b[0]) <= 1e-6 * abs(b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])      # (the loss total is an fp32 atomic sum over blocks)
    assert abs(a[0] - o[0]) < 1e-4 * max(1.0, abs(o[0]))
    assert float((a[1] - o[1]).abs().max()) < 2e-3
    assert rel(a[3], o[3].double()) < 2e-2 and rel(a[4], o[4].double()) < 2e-2
    print(f"pooled scoring: loss {a[0]:.6f} (128 x 128 kernels: {o[0]:.6f}); dP rel {rel(a[3], o[3].double()):.2e}, dE rel {rel(a[4], o[4].double()):.2e}")
