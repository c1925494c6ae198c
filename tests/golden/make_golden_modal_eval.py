#!/usr/bin/env python
"""g17: the MODAL evaluation path captured from the imported reference -- HR@10 / nDCG@10 THROUGH the text tower.

    python tests/golden/make_golden_modal_eval.py          (build container only: needs /root/reference)

``T/data_utils/metrics.py:60-74`` (``get_item_embeddings`` with ``use_modal=True``: every title through
``model.module.bert_encoder`` in eval mode) -> ``:77-107`` (``eval_model``: SASRec user states, full score matrix, history
set to -inf, column 0 dropped, argsort rank, Hit@10 / nDCG@10 per user).  BERT micro tower (the g5 architecture), weights from
``det_param`` (re-created on both sides), synthetic titles and histories.  Stored: inputs (titles, sequences, pop) and the
reference's numeric outputs (item embeddings, per-user hit / nDCG, means, and the per-user score margin between the target and
its nearest competitor, so that the test can tell a rank that legitimately sits within fp32 noise from a wrong one).
No reference source text is stored."""
from __future__ import annotations

import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts the reference package and the repo on sys.path)

from idvs.morec_amd.utils.detgen import det_randint  # noqa: E402


def g17(out):
    import torch.distributed as dist
    from data_utils import metrics as ref_metrics
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1)
    S, D, T, item_num, U = 8, 64, 30, 60, 61
    args = mg.make_args(max_seq_len=S, embedding_dim=D, word_embedding_dim=64)
    pop = mg.zipf_pop(item_num, "pop.g17")
    kw = dict(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
              max_position_embeddings=64)
    m = mg.build_modal(args, kw, item_num, pop)
    content = mg.synth_titles("g17", item_num, T, 512)
    wrap = types.SimpleNamespace(module=m, eval=m.eval, train=m.train)
    eval_seq, hist = {}, {}
    for u in range(U):
        L = int(det_randint(f"g17.len{u}", (1,), 3, S + 2)[0])
        seq = [int(v) for v in det_randint(f"g17.seq{u}", (L,), 1, item_num + 1)]
        eval_seq[u] = seq
        hist[u] = torch.LongTensor(np.array(seq[:-1]))
    captured, captured_print = {}, {}
    orig_concat = ref_metrics.eval_concat

    def cap_concat(eval_list, sampler):
        captured["hit"], captured["ndcg"] = [e.clone().numpy() for e in eval_list]
        return orig_concat(eval_list, sampler)

    ref_metrics.eval_concat = cap_concat
    ref_metrics.print_metrics = lambda x, log, v: captured_print.setdefault("mean", list(x))
    emb = ref_metrics.get_item_embeddings(wrap, content, 16, args, True, "cpu")
    hit10 = ref_metrics.eval_model(wrap, hist, eval_seq, emb, 16, args, item_num, logging.getLogger("g17"), "valid", "cpu")
    # score margins (own arithmetic on the captured embeddings, for the test's benefit): the distance from the target's score to
    # the nearest other unmasked score
    margins = np.zeros(U)
    with torch.no_grad():
        for u in range(U):
            seq = eval_seq[u]
            toks = seq[:-1]
            idx = np.zeros(S, dtype=np.int64)
            lm = np.zeros((1, S), dtype=np.float32)
            idx[S - len(toks):] = toks
            lm[0, S - len(toks):] = 1
            prec = m.user_encoder(emb[torch.from_numpy(idx)][None], torch.from_numpy(lm), "cpu")[0, -1]
            sc = (emb @ prec).numpy().astype(np.float64)
            sc[np.array(toks)] = -np.inf
            t = sc[seq[-1]]
            others = np.delete(sc[1:], seq[-1] - 1)
            margins[u] = np.min(np.abs(others[np.isfinite(others)] - t)) if np.isfinite(t) else np.inf
    res = dict(cfg=np.array([S, D, T, item_num, U]), pop=pop, content=content, hit_per_user=captured["hit"][:U],
               ndcg_per_user=captured["ndcg"][:U], hit10=np.float64(captured_print["mean"][0]),
               ndcg10=np.float64(captured_print["mean"][1]), item_embeddings=emb.numpy(), margins=margins)
    for u in range(U):
        res[f"seq.{u}"] = np.array(eval_seq[u])
    np.savez_compressed(os.path.join(out, "g17_eval_modal.npz"), **res)
    print("g17 done: hit10", hit10, captured_print, "min score margin", margins.min())


if __name__ == "__main__":
    g17(HERE)
