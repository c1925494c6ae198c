"""TrainStep under data parallelism on ONE MI355X: two ``gloo`` ranks sharing ``cuda:0`` (RCCL refuses two ranks on one
device, so the transport is gloo; the bookkeeping under test -- pooled negatives, dE reduce-scatter, bucketed gradient
reduction driven by the backward-pass callbacks, closing sweep -- is transport independent).  N ranks x B must be the
single-process step at batch N*B: same loss, same parameters after two AdamW steps."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(dtype, tower):
    """Deterministic model + batch of 8 users (tower: 'text' | 'id' | 'swin')."""
    from idvs.morec_amd.model import BertShape, HipBertModel, Model
    from idvs.morec_amd.utils.detgen import det_param
    S, D, T, item_num, B = 6, 64, 12, 90, 8
    rng = np.random.default_rng(11)
    pop = rng.random(item_num + 1) + 0.05
    pop[1:] /= pop[1:].sum()
    pop[0] = 1.0
    common = dict(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2, compute_dtype=dtype)
    if tower == "swin":
        from idvs.morec_amd.model.swin import HipSwinForImageClassification
        from idvs.morec_amd.swin_engine import SwinShape
        vshape = SwinShape.named("swin_micro")
        args = types.SimpleNamespace(CV_model_load="swin_micro", **common)
        model = Model(args, item_num, True, HipSwinForImageClassification(vshape, D), pop)
        content = rng.standard_normal((item_num + 1, 3, vshape.image_size, vshape.image_size)).astype(np.float32)
        content[0] = 0
    else:
        shape = BertShape(vocab_size=700, hidden_size=64, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=32)
        args = types.SimpleNamespace(num_words_title=T, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                     bert_model_load="bert_x", word_embedding_dim=64, **common)
        model = Model(args, item_num, tower == "text", HipBertModel(shape) if tower == "text" else None, pop)
        content = np.zeros((item_num + 1, 2 * T), dtype=np.int64)
        for i in range(1, item_num + 1):
            L = int(rng.integers(2, T + 1))
            content[i, :L] = rng.integers(1, 700, L)
            content[i, T:T + L] = 1
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(torch.from_numpy(det_param(k, tuple(v.shape))))
    ids = np.zeros((B, S + 1), dtype=np.int64)
    lm = np.zeros((B, S), dtype=np.float32)
    for b in range(B):
        L = int(rng.integers(2, S + 2))
        ids[b, S + 1 - L:] = rng.integers(1, item_num + 1, L)
        lm[b, S + 1 - L:] = 1
    model.eval()
    return model.to("cuda"), ids, lm, content


def _steps(model, ids, lm, content, tower, pool, n_steps=2):
    from idvs.morec_amd.train_step import TrainStep
    ts = TrainStep(model, lr=1e-3, fine_tune_lr=5e-4, l2_weight=0.01, fine_tune_l2_weight=0.02, pool_negatives=pool)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    flat = ids.reshape(-1)
    items = flat if tower == "id" else content[flat]
    losses = [ts.step(dev(flat), dev(items), dev(lm)) for _ in range(n_steps)]
    return ts, [float(x) for x in losses]


def _worker(rank, world, port, q, dtype, tower, overlap):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MOREC_OVERLAP_REDUCE"] = "1" if overlap else "0"
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, ids, lm, content = _build(dtype, tower)
    B = ids.shape[0] // world
    ts, losses = _steps(model, ids[rank * B:(rank + 1) * B], lm[rank * B:(rank + 1) * B], content, tower, True)
    tot = torch.tensor(losses, dtype=torch.float64)
    dist.all_reduce(tot)                      # the step returns the local share of the pooled loss
    buckets = sorted(str(k) for k in ts.buckets)
    n_reduced = len(ts._reduced)
    if rank == 0:
        q.put((tot.tolist(), {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}, buckets, n_reduced))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("tower,dtype,overlap", [("text", "fp32", True), ("text", "fp32", False), ("text", "bf16", True),
                                                 ("id", "fp32", True), ("swin", "fp32", True)])
def test_two_ranks_equal_single_process(tower, dtype, overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, dtype, tower, overlap)) for r in range(2)]
    for pr in procs:
        pr.start()
    losses2, sd2, buckets, n_reduced = q.get(timeout=500)
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    model, ids, lm, content = _build(dtype, tower)
    _, losses1 = _steps(model, ids, lm, content, tower, True)
    tol_l = 2e-5 if dtype == "fp32" else 2e-2
    assert all(abs(a - b) < tol_l * max(1.0, abs(b)) for a, b in zip(losses2, losses1)), (losses2, losses1)
    # two Adam steps move a weight by <= 2 lr; sign flips of eps-dominated gradient elements may cost a fraction of that
    worst, lr = 0.0, 1e-3
    for k, v in model.state_dict().items():
        if "pooler" in k or k.endswith(("key.bias", "k_proj.bias", "w_K.bias")):
            continue      # key biases: the true gradient is zero (softmax shift invariance), Adam amplifies the rounding noise
        worst = max(worst, float(np.abs(v.detach().float().cpu().numpy() - sd2[k]).max()))
    assert worst < (0.2 * lr if dtype == "fp32" else 2.5 * lr), worst
    if tower == "text":
        assert buckets == sorted(str(("layer", l)) for l in range(3))
        assert n_reduced == (3 + 1 + 1 if overlap else 2)       # 3 layers + head + embeddings sweep / one sweep per group
    if tower == "swin":
        assert len(buckets) >= 1 and n_reduced >= len(buckets) + 1
    print(tower, dtype, "2-rank vs single:", losses2, losses1, "worst param diff", worst, "slices", n_reduced)
