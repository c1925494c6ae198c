"""Pin the CPU oracle against golden vectors captured from the imported reference
(``tests/golden/make_golden.py``).  CPU-only; runs in the default (not gpu) suite."""
import os

import numpy as np
import pytest
import torch

import morec_oracle as orc
from morec_oracle import bookkeeping as bk
from idvs.morec_amd.model.spec import (BM, TE, BertShape, bert_param_shapes, model_param_shapes,
                                        sasrec_param_shapes)
from helpers import det_state, relerr


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


ID_CASES = ["a", "b", "c", "d", "e"]


@pytest.mark.parametrize("case", ID_CASES)
def test_g1_bookkeeping_bit_exact(golden_dir, case):
    g = _load(golden_dir, "g1_g4_id_tower.npz")
    B, S = int(g[f"{case}.B"]), int(g[f"{case}.S"])
    ids, log_mask = g[f"{case}.ids"], g[f"{case}.log_mask"]
    rows = bk.valid_rows(log_mask)
    labels = bk.ce_labels(B, S)
    assert np.array_equal(labels[rows], g[f"{case}.labels_valid"])
    masked = (~bk.column_valid(log_mask))[None, :] | bk.reject_mask(ids, B, S).reshape(B * S, -1)
    assert np.array_equal(masked[rows], g[f"{case}.masked_valid"])
    # collate restatement reproduces the padded rows it was derived from
    for b in range(B):
        seq = ids[b][ids[b] != 0] if (ids[b] != 0).any() else ids[b][-1:]
        L = int(log_mask[b].sum()) + 1
        i2, m2 = bk.collate_train_sample(ids[b][-L:], S)
        assert np.array_equal(i2, ids[b]) and np.array_equal(m2, log_mask[b])


@pytest.mark.parametrize("case", ID_CASES)
def test_g4_id_tower_loss_and_grads(golden_dir, case):
    g = _load(golden_dir, "g1_g4_id_tower.npz")
    B, S, item_num, D = (int(g[f"{case}.{k}"]) for k in ("B", "S", "item_num", "D"))
    shapes = model_param_shapes(max_seq_len=S, embedding_dim=D, n_blocks=2, item_num=item_num, use_modal=False)
    p = {k: v.requires_grad_(True) for k, v in det_state(shapes).items()}
    ids = torch.from_numpy(g[f"{case}.ids"]).view(-1)
    loss, parts = None, None
    score = p["id_embedding.weight"][ids]
    prec = orc.sasrec_forward(p, score.view(B, S + 1, D)[:, :-1], torch.from_numpy(g[f"{case}.log_mask"]), 2)
    loss, parts = orc.inbatch_ce_loss(prec.reshape(-1, D), score, ids, g[f"{case}.log_mask"], g[f"{case}.pop"], S,
                                      return_parts=True)
    assert abs(loss.item() - float(g[f"{case}.loss"])) < 2e-5
    got = parts["logits"][parts["rows"]].detach().numpy()
    assert np.abs(got - g[f"{case}.logits_valid"]).max() < 1e-4
    loss.backward()
    ge = p["id_embedding.weight"].grad.clone()
    ge[0] = 0  # nn.Embedding(padding_idx=0) drops the padding row's gradient (T/model/model.py:27)
    assert np.abs(ge.numpy() - g[f"{case}.grad_id_embedding"]).max() < 2e-6
    for k in [k for k in g.files if k.startswith(f"{case}.grad.")]:
        name = k[len(f"{case}.grad."):]
        assert np.abs(p[name].grad.numpy() - g[k]).max() < 5e-6, name


def test_g9_pooled_equals_single_process(golden_dir):
    """N ranks x B with pooled negatives == reference at batch N*B (rank-major)."""
    g = _load(golden_dir, "g1_g4_id_tower.npz")
    case, N = "e", 4
    Btot, S, item_num, D = (int(g[f"{case}.{k}"]) for k in ("B", "S", "item_num", "D"))
    B = Btot // N
    shapes = model_param_shapes(max_seq_len=S, embedding_dim=D, n_blocks=2, item_num=item_num, use_modal=False)
    p = det_state(shapes)
    ids_all, lm_all = g[f"{case}.ids"], g[f"{case}.log_mask"]
    score_all = p["id_embedding.weight"][torch.from_numpy(ids_all).view(-1)]
    n_valid = int((lm_all != 0).sum())
    total = 0.0
    for r in range(N):
        ids = ids_all[r * B:(r + 1) * B]
        lm = lm_all[r * B:(r + 1) * B]
        score = score_all[r * B * (S + 1):(r + 1) * B * (S + 1)]
        prec = orc.sasrec_forward(p, score.view(B, S + 1, D)[:, :-1], torch.from_numpy(lm), 2).reshape(-1, D)
        total += orc.inbatch_ce_loss(prec, score_all, ids, lm, g[f"{case}.pop"], S, pool_ids=ids_all,
                                     pool_log_mask=lm_all, col_offset=r * B * (S + 1), n_valid_total=n_valid).item()
    assert abs(total - float(g[f"{case}.loss"])) < 2e-5


def test_g2_read_behaviors(golden_dir):
    g = _load(golden_dir, "g2_read_behaviors.npz")
    name_to_id, _ = orc.read_news_ref(os.path.join(golden_dir, "g2_items.tsv"))
    r = orc.read_behaviors_ref(os.path.join(golden_dir, "g2_users.tsv"), name_to_id, int(g["S"]), int(g["min_seq_len"]))
    assert r["item_num"] == int(g["item_num"]) and len(r["users_train"]) == int(g["n_users"])
    assert np.array_equal(r["pop_prob_list"], g["pop"])  # float64, bit-exact
    now = {n: r["before_to_now"][name_to_id[n]] for n in name_to_id if name_to_id[n] in r["before_to_now"]}
    assert now == dict(zip(g["names"].tolist(), g["name_ids"].tolist()))
    for u in range(int(g["n_users"])):
        assert np.array_equal(r["users_train"][u], g[f"train.{u}"])
        assert np.array_equal(r["users_valid"][u], g[f"valid.{u}"])
        assert np.array_equal(r["users_test"][u], g[f"test.{u}"])
        assert np.array_equal(r["hist_valid"][u], g[f"hv.{u}"])
        assert np.array_equal(r["hist_test"][u], g[f"ht.{u}"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_g3_sasrec(golden_dir, case):
    from idvs.morec_amd.utils.detgen import det_normal
    g = _load(golden_dir, "g3_sasrec.npz")
    B, S, D, heads, blocks = (int(v) for v in g[f"{case}.cfg"])
    p = {k: v.requires_grad_(True) for k, v in
         det_state(sasrec_param_shapes(S, D, blocks, prefix="transformer_encoder.")).items()}
    x = torch.from_numpy(det_normal(f"g3{case}.x", (B, S, D), std=0.5)).requires_grad_(True)
    R = torch.from_numpy(det_normal(f"g3{case}.R", (B, S, D), std=1.0))
    y = orc.sasrec_forward(p, x, torch.from_numpy(g[f"{case}.log_mask"]), heads, prefix="transformer_encoder.")
    assert np.abs(y.detach().numpy() - g[f"{case}.y"]).max() < 2e-5
    (y * R).sum().backward()
    assert relerr(x.grad.numpy(), g[f"{case}.dx"]) < 2e-5
    for k in [k for k in g.files if k.startswith(f"{case}.grad.")]:
        assert relerr(p[k[len(f"{case}.grad."):]].grad.numpy(), g[k]) < 5e-5, k


def _modal_state(S, D, item_num, bert):
    shapes = model_param_shapes(max_seq_len=S, embedding_dim=D, n_blocks=2, item_num=item_num, use_modal=True, bert=bert)
    return {k: v.requires_grad_(True) for k, v in det_state(shapes).items()}


def test_g5_bert_micro(golden_dir):
    from idvs.morec_amd.utils.detgen import det_normal
    g = _load(golden_dir, "g5_g8_bert_micro.npz")
    S, D, T, item_num, B = (int(v) for v in g["cfg"])
    bert = BertShape.named("micro")
    p = _modal_state(S, D, item_num, bert)
    items = torch.from_numpy(g["content"][g["ids"].reshape(-1)])
    vec = orc.text_encoder_forward(p, items, bert.num_attention_heads)
    real = g["ids"].reshape(-1) != 0  # all-PAD title rows are implementation-defined (SURVEY.md §8c hazard 1)
    assert np.abs(vec.detach().numpy() - g["item_vecs"])[real].max() < 2e-5
    # padded titles: HF eager (finfo.min additive mask) gives a uniform softmax; the oracle restates that
    assert np.abs(vec.detach().numpy() - g["item_vecs"]).max() < 2e-5
    R = torch.from_numpy(det_normal("g5.R", (B * (S + 1), D)))
    (vec * R).sum().backward()
    for k in [k for k in g.files if k.startswith("enc_grad.")]:
        assert relerr(p[k[len("enc_grad."):]].grad.numpy(), g[k]) < 5e-5, k
    for k in [k for k in g.files if k.startswith("enc_grad_norm.")]:
        name = k[len("enc_grad_norm."):]
        if "pooler" in name:
            continue
        assert abs(p[name].grad.double().norm().item() - float(g[k])) <= 1e-4 * float(g[k]) + 2e-5, name
    for v in p.values():
        v.grad = None
    loss = orc.model_forward(p, torch.from_numpy(g["ids"]).view(-1), items, torch.from_numpy(g["log_mask"]), g["pop"],
                             max_seq_len=S, embedding_dim=D, n_heads=2, use_modal=True,
                             bert_heads=bert.num_attention_heads)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    loss.backward()
    for k in [k for k in g.files if k.startswith("grad_norm.")]:
        name = k[len("grad_norm."):]
        if "pooler" in name:
            continue
        assert abs(p[name].grad.double().norm().item() - float(g[k])) <= 2e-4 * float(g[k]) + 2e-5, name
    # g8: one AdamW step, two param groups (T/run.py:150-162)
    for k in [k for k in g.files if k.startswith("step_delta.")]:
        name = k[len("step_delta."):]
        lr = 5e-5 if "bert_model" in name else 1e-4
        w = p[name].detach().clone()
        before = w.clone()
        orc.adamw_step(w, p[name].grad, torch.zeros_like(w), torch.zeros_like(w), 1, lr, 0.01)
        # first Adam step = -lr*g/(|g|+eps): elements with |g| ~ eps are dominated by rounding noise in g
        big = (p[name].grad.abs() > 1e-5).numpy()
        assert big.mean() > 0.05
        assert np.abs((w - before).numpy() - g[k])[big].max() < 2e-7, name


@pytest.mark.parametrize("name", ["tiny", "base"])
def test_g6_full_size_scalars(golden_dir, name):
    g = _load(golden_dir, "g6_full_scalars.npz")
    S, D, T, item_num, B = (int(v) for v in g[f"{name}.cfg"])
    bert = BertShape.named(name)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    p = _modal_state(S, D, item_num, bert)
    items = torch.from_numpy(g[f"{name}.content"][g[f"{name}.ids"].reshape(-1)])
    with torch.no_grad():
        loss = orc.model_forward(p, torch.from_numpy(g[f"{name}.ids"]).view(-1), items,
                                 torch.from_numpy(g[f"{name}.log_mask"]), g[f"{name}.pop"], max_seq_len=S,
                                 embedding_dim=D, n_heads=2, use_modal=True, bert_heads=bert.num_attention_heads)
    assert abs(loss.item() - float(g[f"{name}.loss"])) < 5e-5


def test_g18_long_sequences(golden_dir):
    """Behaviour sequences of 40 items and texts of 50 tokens (the reference's abstracts / bodies, T/parameters.py:43-44): loss and every
    gradient norm of the oracle against the reference's own numbers (what the 64 x 64 attention kernels are checked against on the GPU)."""
    g = _load(golden_dir, "g18_long_scalars.npz")
    S, D, T, item_num, B = (int(v) for v in g["long.cfg"])
    assert S > 32 and T > 32 and int(g["long.log_mask"].sum(1).max()) == S
    bert = BertShape.named("tiny")
    p = _modal_state(S, D, item_num, bert)
    items = torch.from_numpy(g["long.content"][g["long.ids"].reshape(-1)])
    loss = orc.model_forward(p, torch.from_numpy(g["long.ids"]).view(-1), items, torch.from_numpy(g["long.log_mask"]), g["long.pop"],
                             max_seq_len=S, embedding_dim=D, n_heads=2, use_modal=True, bert_heads=bert.num_attention_heads)
    assert abs(loss.item() - float(g["long.loss"])) < 2e-5
    loss.backward()
    for k in [k for k in g.files if k.startswith("long.grad_norm.")]:
        name = k[len("long.grad_norm."):]
        if "pooler" in name:
            continue
        assert abs(p[name].grad.double().norm().item() - float(g[k])) <= 2e-4 * float(g[k]) + 2e-5, name


def test_g7_eval(golden_dir):
    g = _load(golden_dir, "g7_eval.npz")
    S, D, item_num, U = (int(v) for v in g["cfg"])
    shapes = model_param_shapes(max_seq_len=S, embedding_dim=D, n_blocks=2, item_num=item_num, use_modal=False)
    p = det_state(shapes)
    emb = p["id_embedding.weight"]
    assert np.abs(emb.numpy() - g["item_embeddings"]).max() == 0.0
    ranks = np.zeros(U, dtype=np.int64)
    scores, hists, targets = [], [], []
    for u in range(U):
        seq = g[f"seq.{u}"]
        tokens, target = seq[:-1], int(seq[-1])
        pad = S + 1 - len(seq)
        x = emb[torch.from_numpy(np.concatenate([np.zeros(pad, dtype=np.int64), tokens]))][None]
        lm = torch.tensor([[0.0] * pad + [1.0] * len(tokens)])
        prec = orc.sasrec_forward(p, x, lm, 2)[:, -1]
        scores.append((prec @ emb.t())[0].numpy())
        hists.append(tokens)
        targets.append(target)
    ranks = orc.eval_ranks(np.stack(scores), hists, np.asarray(targets))
    hit = (ranks <= 10).astype(np.float32)
    ndcg = np.where(ranks <= 10, 1.0 / np.log2(ranks + 1.0), 0.0)
    assert np.array_equal(hit, g["hit_per_user"])
    assert np.abs(ndcg - g["ndcg_per_user"]).max() < 1e-6
    h, n = orc.hit_ndcg_at_k(ranks)
    assert abs(h - float(g["hit10"])) < 1e-6 and abs(n - float(g["ndcg10"])) < 1e-6


def test_g17_modal_eval(golden_dir):
    """HR@10 / nDCG@10 THROUGH the text tower (``T/data_utils/metrics.py:60-74`` with ``use_modal=True`` -> ``:77-107``): the
    oracle's BERT item vectors, SASRec user states and rank bookkeeping against the values captured from the reference."""
    g = _load(golden_dir, "g17_eval_modal.npz")
    S, D, T, item_num, U = (int(v) for v in g["cfg"])
    bert = BertShape.named("micro")
    shapes = model_param_shapes(max_seq_len=S, embedding_dim=D, n_blocks=2, item_num=item_num, use_modal=True, bert=bert)
    p = det_state(shapes)
    with torch.no_grad():
        emb = orc.text_encoder_forward(p, torch.from_numpy(g["content"]), bert.num_attention_heads)
    real = np.arange(item_num + 1) != 0      # row 0 = the all-[PAD] padding item: implementation-defined vector (SURVEY §8c hazard 1)
    assert np.abs(emb.numpy()[real] - g["item_embeddings"][real]).max() < 2e-6
    emb = torch.from_numpy(g["item_embeddings"])     # ranks from the reference's own vectors (row 0 is masked / dropped anyway)
    scores, hists, targets = [], [], []
    with torch.no_grad():
        for u in range(U):
            seq = g[f"seq.{u}"]
            tokens, target = seq[:-1], int(seq[-1])
            pad = S + 1 - len(seq)
            x = emb[torch.from_numpy(np.concatenate([np.zeros(pad, dtype=np.int64), tokens]))][None]
            lm = torch.tensor([[0.0] * pad + [1.0] * len(tokens)])
            prec = orc.sasrec_forward(p, x, lm, 2)[:, -1]
            scores.append((prec @ emb.t())[0].numpy())
            hists.append(tokens)
            targets.append(target)
    ranks = orc.eval_ranks(np.stack(scores), hists, np.asarray(targets))
    hit = (ranks <= 10).astype(np.float32)
    ndcg = np.where(ranks <= 10, 1.0 / np.log2(ranks + 1.0), 0.0)
    assert np.array_equal(hit, g["hit_per_user"])
    assert np.abs(ndcg - g["ndcg_per_user"]).max() < 1e-6
    h, n = orc.hit_ndcg_at_k(ranks)
    assert abs(h - float(g["hit10"])) < 1e-6 and abs(n - float(g["ndcg10"])) < 1e-6
