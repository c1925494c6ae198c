"""CPU: the bucket bookkeeping of the overlapped gradient reduction (``TrainStep._bucket_plan`` / ``_on_ready`` /
``reduce_gradients``).  Whatever order the backward pass reports its milestones in, every element of every gradient arena
must be reduced exactly once.  (The arithmetic itself is covered on the GPU by ``test_train_step_ddp_gpu.py``.)"""
import types

import numpy as np
import pytest
import torch


def _train_step(tower):
    from idvs.morec_amd.model import BertShape, HipBertModel, Model
    from idvs.morec_amd.train_step import TrainStep
    pop = np.ones(101) / 100
    pop[0] = 1.0
    common = dict(max_seq_len=6, embedding_dim=64, num_attention_heads=2, drop_rate=0.0, transformer_block=2, compute_dtype="fp32")
    if tower == "swin":
        from idvs.morec_amd.model.swin import HipSwinForImageClassification
        from idvs.morec_amd.swin_engine import SwinShape
        args = types.SimpleNamespace(CV_model_load="swin_micro", **common)
        model = Model(args, 100, True, HipSwinForImageClassification(SwinShape.named("swin_micro"), 64), pop)
    else:
        shape = BertShape(vocab_size=300, hidden_size=64, num_hidden_layers=3, num_attention_heads=2, intermediate_size=128,
                          max_position_embeddings=32)
        args = types.SimpleNamespace(num_words_title=12, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                     bert_model_load="bert_x", word_embedding_dim=64, **common)
        model = Model(args, 100, tower == "text", HipBertModel(shape) if tower == "text" else None, pop)
    return TrainStep(model, lr=1e-3, fine_tune_lr=1e-4, l2_weight=0.0, fine_tune_l2_weight=0.0)


@pytest.mark.parametrize("tower,order", [("text", "backward"), ("text", "shuffled"), ("text", "none"), ("id", "none"),
                                         ("swin", "backward")])
def test_every_gradient_element_reduced_once(tower, order):
    ts = _train_step(tower)
    ts.world, ts.collectives = 2, True             # pretend: the slices are recorded instead of sent
    calls = []
    ts._reduce_slice = lambda gi, lo, hi: (calls.append((gi, lo, hi)), ts._reduced.append((gi, lo, hi)))
    ts._pending, ts._reduced = [], []
    keys = sorted(ts.buckets, key=lambda k: -k[1])          # the engines report the last layer / stage first
    if tower == "text":
        assert keys == [("layer", 2), ("layer", 1), ("layer", 0)]
    if tower == "swin":
        assert keys and all(k[0] == "stage" for k in keys)
    if order == "shuffled":
        keys = [keys[1], keys[0], keys[2]]
    if order != "none":
        if tower == "text":
            ts._on_ready("head")
        for k in keys:
            ts._on_ready(k)
        ts._on_ready(("layer", 99))                # unknown milestones are ignored
    ts.reduce_gradients()
    for gi, grp in enumerate(ts.groups):
        cover = torch.zeros(grp["arena"].numel, dtype=torch.int32)
        for g_, lo, hi in calls:
            if g_ == gi:
                assert 0 <= lo < hi <= grp["arena"].numel
                cover[lo:hi] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1, (tower, order, gi)
    if tower == "text" and order != "none":
        assert len(calls) == 1 + 3 + 1             # recommender arena, 3 layers, the embeddings sweep


def test_bucket_slices_hold_exactly_their_layer():
    ts = _train_step("text")
    a0 = ts.groups[0]["arena"]
    for (tag, l), (lo, hi) in ts.buckets.items():
        inside = [n for n, (o, cnt, _) in a0.offsets.items() if lo <= o < hi]
        assert inside and all(f".encoder.layer.{l}." in n for n in inside)
        outside = [n for n, (o, cnt, _) in a0.offsets.items() if not (lo <= o < hi)]
        assert all(f".encoder.layer.{l}." not in n for n in outside)
