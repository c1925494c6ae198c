#!/usr/bin/env python
"""Capture golden vectors from the IMPORTED reference (westlake-repl/IDvs.MoRec @ /root/reference).

Run ONCE in the build container (the only place /root/reference exists):

    python tests/golden/make_golden.py

It imports the reference's ``inbatch_sasrec_e2e_text`` package unmodified (CPU, fp32,
``local_rank='cpu'``, gloo world_size 1 for the eval path), feeds it synthetic inputs and
deterministically generated weights (``idvs.morec_amd.utils.detgen.det_param`` -- both sides can
re-create them, so only inputs and numeric OUTPUTS are stored) and writes small ``.npz`` fixtures
next to this file.  No reference source text is stored anywhere in this repository.

Fixture map: g1 bookkeeping (ints, bit-exact) + ID-tower loss; g2 read_behaviors;
g3 SASRec micro fwd/bwd; g4 ID tower loss + grads; g5 BERT micro item vectors + grads; g6
BERT-tiny / BERT-base scalars; g7 eval HR@10/nDCG@10; g8 one AdamW step; g9 pooled (N*B) goldens.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/inbatch_sasrec_e2e_text"
sys.path.insert(0, REF)

from idvs.morec_amd.utils.detgen import det_param, det_normal, det_randint, det_uniform  # noqa: E402

from model import Model as RefModel  # noqa: E402  (reference package)
from model.encoders import User_Encoder as RefUserEncoder  # noqa: E402

torch.set_num_threads(8)


def make_args(**kw):
    d = dict(max_seq_len=20, embedding_dim=64, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
             num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
             bert_model_load="bert_micro", word_embedding_dim=64, num_workers=0)
    d.update(kw)
    return types.SimpleNamespace(**d)


def load_det(module: torch.nn.Module, seed: int = 12345):
    sd = module.state_dict()
    new = {k: torch.from_numpy(det_param(k, tuple(v.shape), seed=seed)) for k, v in sd.items()
           if v.dtype.is_floating_point}
    module.load_state_dict(new, strict=False)
    return module


class Capture(torch.nn.Module):
    """Stands in for ``Model.criterion`` to expose what the reference feeds its CrossEntropyLoss."""

    def __init__(self):
        super().__init__()
        self.ce = torch.nn.CrossEntropyLoss()

    def forward(self, logits, labels):
        self.logits, self.labels = logits.detach().clone(), labels.detach().clone()
        return self.ce(logits, labels)


def synth_batch(name, B, S, item_num, ragged=True, collide=True, seed=1):
    """ids int64[B,S+1] left-padded, log_mask f32[B,S].  Adds intra-user repeats and cross-user collisions."""
    ids = np.zeros((B, S + 1), dtype=np.int64)
    log_mask = np.zeros((B, S), dtype=np.float32)
    for b in range(B):
        L = S + 1 if not ragged else int(det_randint(f"{name}.len{b}", (1,), 3, S + 2, seed=seed)[0])
        seq = det_randint(f"{name}.seq{b}", (L,), 1, item_num + 1, seed=seed)
        if collide and L >= 4:
            seq[-2] = seq[0]                      # intra-user repeat
            if b > 0:
                seq[1] = ids[b - 1, -1]           # cross-user collision with previous user's last item
        ids[b, S + 1 - L:] = seq
        log_mask[b, S + 1 - L:S] = 1.0
        log_mask[b, S - (L - 1):] = 1.0
    return ids, log_mask


def zipf_pop(item_num, name):
    w = 1.0 / np.arange(1, item_num + 1, dtype=np.float64)
    perm = np.argsort(det_uniform(name, (item_num,)))
    w = w[perm]
    return np.append([1.0], w / w.sum())


# ------------------------------------------------------------------------------------------
def g1_g4_g9(out):
    res = {}
    cases = [("a", 6, 5, 40, True), ("b", 3, 20, 300, False), ("c", 1, 20, 300, True), ("d", 8, 10, 25, True),
             ("e", 16, 20, 300, True)]
    for name, B, S, item_num, ragged in cases:
        D = 64
        args = make_args(max_seq_len=S, embedding_dim=D)
        pop = zipf_pop(item_num, f"pop.{name}")
        torch.manual_seed(0)
        m = RefModel(args, item_num, False, None, pop)
        load_det(m)
        m.eval()
        cap = Capture()
        m.criterion = cap
        ids, log_mask = synth_batch(f"g1{name}", B, S, item_num, ragged=ragged)
        ids_t = torch.from_numpy(ids).view(-1)
        m.zero_grad()
        loss = m(ids_t, ids_t.clone(), torch.from_numpy(log_mask), "cpu")
        loss.backward()
        res[f"{name}.B"], res[f"{name}.S"], res[f"{name}.item_num"], res[f"{name}.D"] = B, S, item_num, D
        res[f"{name}.ids"], res[f"{name}.log_mask"], res[f"{name}.pop"] = ids, log_mask, pop
        res[f"{name}.loss"] = np.float32(loss.item())
        res[f"{name}.labels_valid"] = cap.labels.numpy()
        res[f"{name}.masked_valid"] = (cap.logits == -1e4).numpy()
        res[f"{name}.logits_valid"] = cap.logits.numpy().astype(np.float32)
        res[f"{name}.grad_id_embedding"] = m.id_embedding.weight.grad.numpy()
        g = dict(m.named_parameters())
        for k in ["user_encoder.transformer_encoder.transformer_blocks.0.multi_head_attention.w_Q.weight",
                  "user_encoder.transformer_encoder.transformer_blocks.1.feed_forward.w_2.weight",
                  "user_encoder.transformer_encoder.position_embedding.weight",
                  "user_encoder.transformer_encoder.layer_norm.weight"]:
            res[f"{name}.grad.{k}"] = g[k].grad.numpy()
    np.savez_compressed(os.path.join(out, "g1_g4_id_tower.npz"), **res)
    print("g1/g4/g9 done", {k: float(v) for k, v in res.items() if k.endswith(".loss")})


def g2(out):
    import logging
    from data_utils.preprocess import read_behaviors, read_news
    n_items, n_users, S = 60, 40, 8
    news = os.path.join(out, "g2_items.tsv")
    beh = os.path.join(out, "g2_users.tsv")
    with open(news, "w") as f:
        for i in range(n_items):
            f.write(f"N{i}\tt\ta\n")
    with open(beh, "w") as f:
        for u in range(n_users):
            L = int(det_randint(f"g2.len{u}", (1,), 3, 16)[0])
            seq = det_randint(f"g2.seq{u}", (L,), 0, n_items - 8)   # last 8 items never occur
            f.write(f"U{u}\t" + " ".join(f"N{int(i)}" for i in seq) + "\n")
    a, b, c = read_news(news)
    log = logging.getLogger("g2")
    item_num, item_id_to_dic, tr, va, te, hv, ht, name2id, pop = read_behaviors(beh, a, b, c, S, 5, log)
    res = dict(item_num=item_num, S=S, min_seq_len=5, pop=np.asarray(pop, dtype=np.float64),
               n_users=len(tr),
               names=np.array(sorted(name2id, key=lambda k: name2id[k])),
               name_ids=np.array([name2id[k] for k in sorted(name2id, key=lambda k: name2id[k])]))
    for u in tr:
        res[f"train.{u}"], res[f"valid.{u}"], res[f"test.{u}"] = np.array(tr[u]), np.array(va[u]), np.array(te[u])
        res[f"hv.{u}"], res[f"ht.{u}"] = hv[u].numpy(), ht[u].numpy()
    np.savez_compressed(os.path.join(out, "g2_read_behaviors.npz"), **res)
    print("g2 done: item_num", item_num, "users", len(tr))


def g3(out):
    res = {}
    for name, B, S, D, heads, blocks in [("a", 5, 8, 64, 2, 2), ("b", 3, 20, 128, 2, 1)]:
        enc = RefUserEncoder(item_num=10, max_seq_len=S, item_dim=D, num_attention_heads=heads, dropout=0.0,
                             n_layers=blocks)
        load_det(enc)
        enc.eval()
        x = torch.from_numpy(det_normal(f"g3{name}.x", (B, S, D), std=0.5)).requires_grad_(True)
        _, log_mask = synth_batch(f"g3{name}", B, S, 50, ragged=True, collide=False)
        R = torch.from_numpy(det_normal(f"g3{name}.R", (B, S, D), std=1.0))
        y = enc(x, torch.from_numpy(log_mask), "cpu")
        (y * R).sum().backward()
        res[f"{name}.cfg"] = np.array([B, S, D, heads, blocks])
        res[f"{name}.log_mask"] = log_mask
        res[f"{name}.y"] = y.detach().numpy()
        res[f"{name}.dx"] = x.grad.numpy()
        for k, p in enc.named_parameters():
            res[f"{name}.grad.{k}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(out, "g3_sasrec.npz"), **res)
    print("g3 done")


def build_modal(args, bert_cfg_kw, item_num, pop):
    from transformers import BertConfig, BertModel
    cfg = BertConfig(attn_implementation="eager", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                     **bert_cfg_kw)
    bert = BertModel(cfg)
    m = RefModel(args, item_num, True, bert, pop)
    load_det(m)
    m.eval()
    return m


def synth_titles(name, item_num, T, vocab):
    """item_content int64[item_num+1, 2T]: [CLS] tokens [SEP] PAD..., row 0 all zero (``preprocess.py:135-136``)."""
    content = np.zeros((item_num + 1, 2 * T), dtype=np.int64)
    for i in range(1, item_num + 1):
        L = int(det_randint(f"{name}.tl{i}", (1,), 4, T + 1)[0])
        toks = det_randint(f"{name}.tt{i}", (L,), 5, vocab)
        toks[0], toks[-1] = 1, 2
        content[i, :L] = toks
        content[i, T:T + L] = 1
    return content


def g5_g8(out):
    res = {}
    S, D, T, item_num, B = 6, 64, 30, 50, 4
    args = make_args(max_seq_len=S, embedding_dim=D, word_embedding_dim=64)
    pop = zipf_pop(item_num, "pop.g5")
    kw = dict(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
              max_position_embeddings=64)
    m = build_modal(args, kw, item_num, pop)
    content = synth_titles("g5", item_num, T, 512)
    ids, log_mask = synth_batch("g5", B, S, item_num, ragged=True)
    items = torch.from_numpy(content[ids.reshape(-1)])
    # item vectors + grads through the encoder alone
    R = torch.from_numpy(det_normal("g5.R", (B * (S + 1), D)))
    m.zero_grad()
    vec = m.bert_encoder(items)
    (vec * R).sum().backward()
    res["cfg"] = np.array([S, D, T, item_num, B])
    res["content"], res["ids"], res["log_mask"], res["pop"] = content, ids, log_mask, pop
    res["item_vecs"] = vec.detach().numpy()
    named = dict(m.named_parameters())
    for k, p in named.items():
        if p.grad is not None:
            res[f"enc_grad_norm.{k}"] = np.float64(p.grad.double().norm().item())
    for k in ["bert_encoder.text_encoders.title.fc.weight",
              "bert_encoder.text_encoders.title.bert_model.encoder.layer.0.attention.self.query.weight",
              "bert_encoder.text_encoders.title.bert_model.encoder.layer.1.output.dense.bias",
              "bert_encoder.text_encoders.title.bert_model.embeddings.position_embeddings.weight",
              "bert_encoder.text_encoders.title.bert_model.embeddings.LayerNorm.weight"]:
        res[f"enc_grad.{k}"] = named[k].grad.numpy().copy()
    # full model loss + grads
    m.zero_grad()
    loss = m(torch.from_numpy(ids).view(-1), items, torch.from_numpy(log_mask), "cpu")
    loss.backward()
    res["loss"] = np.float32(loss.item())
    for k, p in named.items():
        if p.grad is not None:
            res[f"grad_norm.{k}"] = np.float64(p.grad.double().norm().item())
    # g8: one AdamW step with the two param groups of T/run.py:150-162 (pooler frozen, no GradScaler)
    pool = ["bert_encoder.text_encoders.title.bert_model.pooler.dense.weight",
            "bert_encoder.text_encoders.title.bert_model.pooler.dense.bias"]
    bert_p = [p for k, p in named.items() if "bert_model" in k and k not in pool]
    rec_p = [p for k, p in named.items() if "bert_model" not in k]
    opt = torch.optim.AdamW([{"params": bert_p, "lr": 5e-5, "weight_decay": 0.01},
                             {"params": rec_p, "lr": 1e-4, "weight_decay": 0.01}])
    before = {k: p.detach().clone() for k, p in named.items()}
    opt.step()
    for k in ["bert_encoder.text_encoders.title.fc.weight",
              "bert_encoder.text_encoders.title.bert_model.encoder.layer.0.attention.self.query.weight",
              "bert_encoder.text_encoders.title.bert_model.embeddings.word_embeddings.weight",
              "user_encoder.transformer_encoder.transformer_blocks.0.feed_forward.w_1.weight",
              "user_encoder.transformer_encoder.layer_norm.bias"]:
        res[f"step_delta.{k}"] = (named[k].detach() - before[k]).numpy()
    opt.zero_grad()
    loss2 = m(torch.from_numpy(ids).view(-1), items, torch.from_numpy(log_mask), "cpu")
    res["loss_after_step"] = np.float32(loss2.item())
    np.savez_compressed(os.path.join(out, "g5_g8_bert_micro.npz"), **res)
    print("g5/g8 done: loss", loss.item(), "->", loss2.item())


def g6(out):
    from transformers import BertConfig
    res = {}
    cfg_base = BertConfig.from_pretrained("/root/reference/pretrained_models/bert_base_uncased").to_dict()
    base_kw = {k: cfg_base[k] for k in ["vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                        "intermediate_size", "max_position_embeddings"]}
    tiny_kw = dict(vocab_size=30522, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                   intermediate_size=512, max_position_embeddings=512)
    for name, kw, B, wdim in [("tiny", tiny_kw, 8, 128), ("base", base_kw, 2, 768)]:
        S, D, T, item_num = 20, 512, 30, 400
        args = make_args(max_seq_len=S, embedding_dim=D, word_embedding_dim=wdim)
        pop = zipf_pop(item_num, f"pop.g6{name}")
        m = build_modal(args, kw, item_num, pop)
        content = synth_titles(f"g6{name}", item_num, T, 30522)
        ids, log_mask = synth_batch(f"g6{name}", B, S, item_num, ragged=True)
        items = torch.from_numpy(content[ids.reshape(-1)])
        m.zero_grad()
        loss = m(torch.from_numpy(ids).view(-1), items, torch.from_numpy(log_mask), "cpu")
        loss.backward()
        res[f"{name}.cfg"] = np.array([S, D, T, item_num, B])
        res[f"{name}.content"], res[f"{name}.ids"], res[f"{name}.log_mask"], res[f"{name}.pop"] = content, ids, log_mask, pop
        res[f"{name}.loss"] = np.float32(loss.item())
        with torch.no_grad():
            vec = m.bert_encoder(items)
        res[f"{name}.item_vec_probe"] = vec[:, :8].numpy()
        for k, p in m.named_parameters():
            if p.grad is not None:
                res[f"{name}.grad_norm.{k}"] = np.float64(p.grad.double().norm().item())
        print("g6", name, "loss", loss.item())
    np.savez_compressed(os.path.join(out, "g6_full_scalars.npz"), **res)


def g18(out):
    """Inputs longer than 32 positions on both towers: behaviour sequences of 40 items (--max_seq_len 40) and texts of 50 tokens (the
    reference's abstracts / bodies: T/parameters.py:43-44, news_attributes) through the reference's own Model -- what the 64 x 64 form of the
    attention kernels is pinned to.  BERT-tiny shape, D = 128."""
    res = {}
    tiny_kw = dict(vocab_size=30522, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                   intermediate_size=512, max_position_embeddings=512)
    name, B, wdim = "long", 6, 128
    S, D, T, item_num = 40, 128, 50, 300
    args = make_args(max_seq_len=S, embedding_dim=D, word_embedding_dim=wdim, num_words_title=T)
    pop = zipf_pop(item_num, "pop.g18")
    m = build_modal(args, tiny_kw, item_num, pop)
    content = synth_titles("g18", item_num, T, 30522)
    ids, log_mask = synth_batch("g18", B, S, item_num, ragged=True)
    ids[0] = det_randint("g18.full", (S + 1,), 1, item_num + 1)          # one user with a FULL history: all 40 rows of the score tile are live
    log_mask[0] = 1.0
    items = torch.from_numpy(content[ids.reshape(-1)])
    m.zero_grad()
    loss = m(torch.from_numpy(ids).view(-1), items, torch.from_numpy(log_mask), "cpu")
    loss.backward()
    res[f"{name}.cfg"] = np.array([S, D, T, item_num, B])
    res[f"{name}.content"], res[f"{name}.ids"], res[f"{name}.log_mask"], res[f"{name}.pop"] = content, ids, log_mask, pop
    res[f"{name}.loss"] = np.float32(loss.item())
    with torch.no_grad():
        vec = m.bert_encoder(items)
    res[f"{name}.item_vec_probe"] = vec[:, :8].numpy()
    for k, p in m.named_parameters():
        if p.grad is not None:
            res[f"{name}.grad_norm.{k}"] = np.float64(p.grad.double().norm().item())
    print("g18 long loss", loss.item(), "titles of", int((content[1:, T:] != 0).sum(1).max()), "tokens at most,", int(log_mask.sum(1).max()), "behaviours at most")
    np.savez_compressed(os.path.join(out, "g18_long_scalars.npz"), **res)


def g19(out):
    """Two text attributes per item -- title (30 tokens) and abstract (50 tokens), ``--news_attributes title,abstract`` -- through the
    reference's Bert_Encoder (T/model/encoders.py:76-117: each attribute through the SAME Text_Encoder, item vector = their mean)."""
    res = {}
    tiny_kw = dict(vocab_size=30522, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                   intermediate_size=512, max_position_embeddings=512)
    B, S, D, Tt, Ta, item_num = 6, 20, 128, 30, 50, 300
    args = make_args(max_seq_len=S, embedding_dim=D, word_embedding_dim=128, news_attributes=["title", "abstract"],
                     num_words_title=Tt, num_words_abstract=Ta)
    pop = zipf_pop(item_num, "pop.g19")
    m = build_modal(args, tiny_kw, item_num, pop)
    content = np.concatenate([synth_titles("g19t", item_num, Tt, 30522), synth_titles("g19a", item_num, Ta, 30522)], axis=1)   # [title ids | title mask | abstract ids | abstract mask]
    ids, log_mask = synth_batch("g19", B, S, item_num, ragged=True)
    items = torch.from_numpy(content[ids.reshape(-1)])
    m.zero_grad()
    loss = m(torch.from_numpy(ids).view(-1), items, torch.from_numpy(log_mask), "cpu")
    loss.backward()
    res["two.cfg"] = np.array([S, D, Tt, Ta, item_num, B])
    res["two.content"], res["two.ids"], res["two.log_mask"], res["two.pop"] = content, ids, log_mask, pop
    res["two.loss"] = np.float32(loss.item())
    with torch.no_grad():
        vec = m.bert_encoder(items)
    res["two.item_vec_probe"] = vec[:, :8].numpy()
    for k, p in m.named_parameters():
        if p.grad is not None:
            res[f"two.grad_norm.{k}"] = np.float64(p.grad.double().norm().item())
    print("g19 two attributes: loss", loss.item(), "row width", content.shape[1])
    np.savez_compressed(os.path.join(out, "g19_two_attributes.npz"), **res)


def g7(out):
    import logging
    import torch.distributed as dist
    from data_utils import metrics as ref_metrics
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1)
    S, D, item_num, U = 8, 64, 120, 37
    args = make_args(max_seq_len=S, embedding_dim=D)
    pop = zipf_pop(item_num, "pop.g7")
    m = RefModel(args, item_num, False, None, pop)
    load_det(m)
    m.eval()
    wrap = types.SimpleNamespace(module=m, eval=m.eval, train=m.train)
    eval_seq, hist = {}, {}
    for u in range(U):
        L = int(det_randint(f"g7.len{u}", (1,), 3, S + 2)[0])
        seq = [int(v) for v in det_randint(f"g7.seq{u}", (L,), 1, item_num + 1)]
        eval_seq[u] = seq
        hist[u] = torch.LongTensor(np.array(seq[:-1]))
    captured = {}
    orig_concat = ref_metrics.eval_concat

    def cap_concat(eval_list, sampler):
        captured["hit"], captured["ndcg"] = [e.clone().numpy() for e in eval_list]
        return orig_concat(eval_list, sampler)

    ref_metrics.eval_concat = cap_concat
    captured_print = {}
    ref_metrics.print_metrics = lambda x, log, v: captured_print.setdefault("mean", list(x))
    item_content = np.arange(item_num + 1)
    emb = ref_metrics.get_item_embeddings(wrap, item_content, 16, args, False, "cpu")
    hit10 = ref_metrics.eval_model(wrap, hist, eval_seq, emb, 16, args, item_num, logging.getLogger("g7"), "valid", "cpu")
    res = dict(cfg=np.array([S, D, item_num, U]), pop=pop, hit_per_user=captured["hit"][:U],
               ndcg_per_user=captured["ndcg"][:U], hit10=np.float64(captured_print["mean"][0]),
               ndcg10=np.float64(captured_print["mean"][1]), item_embeddings=emb.numpy())
    for u in range(U):
        res[f"seq.{u}"] = np.array(eval_seq[u])
    np.savez_compressed(os.path.join(out, "g7_eval.npz"), **res)
    print("g7 done: hit10", hit10, captured_print)


def g10_keys(out):
    """Ordered ``state_dict`` keys + shapes of the reference ``Model`` (ID tower, BERT micro / base with the installed HF
    BertModel): the checkpoint / optimizer-grouping surface."""
    import json
    from transformers import BertConfig, BertModel
    res = {}
    args = make_args(max_seq_len=20, embedding_dim=512, word_embedding_dim=768)
    m = RefModel(args, 100, False, None, [1.0] * 101)
    res["id"] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    cfg = BertConfig.from_pretrained("/root/reference/pretrained_models/bert_base_uncased", attn_implementation="eager")
    cfg.num_hidden_layers = 2      # key pattern per layer is what matters; keeps the fixture small
    m = RefModel(args, 100, True, BertModel(cfg), [1.0] * 101)
    res["modal_base_2layers"] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    res["named_parameters_modal"] = [k for k, _ in m.named_parameters()]
    with open(os.path.join(out, "g10_state_dict_keys.json"), "w") as f:
        json.dump(res, f)
    print("g10 done:", len(res["id"]), len(res["modal_base_2layers"]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = dict(g1=g1_g4_g9, g2=g2, g3=g3, g5=g5_g8, g6=g6, g7=g7, g10=g10_keys, g18=g18, g19=g19)
    for k, fn in todo.items():
        if a.only and k not in a.only.split(","):
            continue
        fn(HERE)
