"""CPU, world_size 2 over gloo: the pooled-negative data-parallel glue of ``functional.InBatchCEFn`` (all-gather of
item vectors / ids / log-pop / validity, global valid-row count, reduce-scatter of dE, column offsets) makes
N ranks x B arithmetically equal to the reference at batch N*B (golden captured from the reference, case 'e').
The LOCAL tile arithmetic is injected from the CPU oracle here (tests only); the product passes ``engine`` (HIP)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "g1_g4_id_tower.npz")


class OracleCE:
    """ce_forward / ce_backward with the same contract as ``engine`` but computed by the CPU oracle's bookkeeping."""

    @staticmethod
    def _loss_sum(ci, P, E):
        from morec_oracle import bookkeeping as bk
        logits = P @ E.t() - ci.col_logpop[None, :]
        rej = torch.from_numpy(bk.reject_mask(ci.row_ids.numpy(), ci.B, ci.S, pool_ids=ci.col_ids.numpy(),
                                              col_offset=ci.col_offset)).view(ci.B * ci.S, -1)
        masked = (ci.col_valid == 0)[None, :] | rej
        logits = torch.where(masked, torch.tensor(-1e4, dtype=logits.dtype), logits)
        labels = torch.from_numpy(bk.ce_labels(ci.B, ci.S) + ci.col_offset)
        lsm = torch.log_softmax(logits, -1)
        row = -lsm[torch.arange(ci.B * ci.S), labels]
        return (row * (ci.row_valid != 0)).sum()

    def ce_forward(self, ci, P, E):
        with torch.no_grad():
            return self._loss_sum(ci, P, E).reshape(1), None

    def ce_backward(self, ci, P, E, saved, gscale_dev, gscale):
        with torch.enable_grad():   # Function.backward runs with grad mode off
            Pg, Eg = P.detach().clone().requires_grad_(True), E.detach().clone().requires_grad_(True)
            (self._loss_sum(ci, Pg, Eg) * gscale * gscale_dev[0]).backward()
        return Pg.grad, Eg.grad


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import morec_oracle as orc
    from idvs.morec_amd import engine
    from idvs.morec_amd.functional import InBatchCEFn
    from idvs.morec_amd.model.spec import model_param_shapes
    from idvs.morec_amd.utils.detgen import det_param
    g = np.load(GOLD)
    case = "e"
    Bt, S, item_num, D = (int(g[f"{case}.{k}"]) for k in ("B", "S", "item_num", "D"))
    B = Bt // world
    shapes = model_param_shapes(max_seq_len=S, embedding_dim=D, n_blocks=2, item_num=item_num, use_modal=False)
    p = {k: torch.from_numpy(det_param(k, s)).double() for k, s in shapes.items()}
    ids = torch.from_numpy(g[f"{case}.ids"][rank * B:(rank + 1) * B])
    lm = torch.from_numpy(g[f"{case}.log_mask"][rank * B:(rank + 1) * B])
    table = p["id_embedding.weight"].clone().requires_grad_(True)
    E = table[ids.view(-1)]
    prec = orc.sasrec_forward(p, E.view(B, S + 1, D)[:, :-1], lm.double(), 2).reshape(-1, D)
    log_pop = torch.log(torch.from_numpy(g[f"{case}.pop"]).float()).double()
    ci = engine.ce_inputs_local(ids.view(-1), lm, log_pop)
    loss = InBatchCEFn.apply(prec, E, ci, True, 1.0, OracleCE())        # local share: sum-reduce convention
    loss.backward()
    total = loss.detach().clone()
    dist.all_reduce(total)
    grad = table.grad.clone()
    grad[0] = 0
    dist.all_reduce(grad)                                                  # SUM of parameter gradients over ranks
    if rank == 0:
        q.put((total.item(), grad.numpy()))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_pooled_negatives_two_ranks_equal_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 400
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    total, grad = q.get(timeout=240)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    g = np.load(GOLD)
    assert abs(total - float(g["e.loss"])) < 2e-5                          # N x B == reference at batch N*B
    ref = g["e.grad_id_embedding"]
    assert np.abs(grad - ref).max() < 5e-6
