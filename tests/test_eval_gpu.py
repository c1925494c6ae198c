"""Evaluation path on a real MI355X: HR@10 / nDCG@10 against the golden captured from the reference's
``eval_model`` (per-user hits and nDCG bit/1e-6 exact -- integer rank bookkeeping), and the run.py driver end to end."""
import logging
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_g7_eval_golden(golden_dir):
    from idvs.morec_amd.data_utils import eval_model, get_item_embeddings
    from idvs.morec_amd.data_utils.metrics import eval_ranks, metrics_from_ranks
    from idvs.morec_amd.model import Model
    from idvs.morec_amd.utils.detgen import det_param
    g = np.load(os.path.join(golden_dir, "g7_eval.npz"))
    S, D, item_num, U = (int(v) for v in g["cfg"])
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                 bert_model_load="x", word_embedding_dim=64, compute_dtype="fp32", num_workers=0)
    m = Model(args, item_num, False, None, g["pop"])
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(torch.from_numpy(det_param(k, tuple(v.shape))))
    m = m.to(DEV)
    eval_seq = {u: [int(v) for v in g[f"seq.{u}"]] for u in range(U)}
    hist = {u: torch.LongTensor(eval_seq[u][:-1]) for u in range(U)}
    emb = get_item_embeddings(m, np.arange(item_num + 1), 16, args, False, DEV)
    assert np.abs(emb.cpu().numpy() - g["item_embeddings"]).max() == 0.0
    ranks = eval_ranks(m, hist, eval_seq, emb, list(range(U)), args, DEV)
    hit, ndcg = metrics_from_ranks(ranks)
    assert np.array_equal(hit.cpu().numpy(), g["hit_per_user"])
    assert np.abs(ndcg.cpu().numpy() - g["ndcg_per_user"]).max() < 1e-6
    hit10 = eval_model(m, hist, eval_seq, emb, 16, args, item_num, logging.getLogger("t"), "valid", DEV)
    assert abs(hit10 - float(g["hit10"])) < 1e-6


def test_g17_modal_eval_golden(golden_dir):
    """HR@10 / nDCG@10 through the BERT tower end to end on the device (golden g17, captured from the reference's
    ``get_item_embeddings(use_modal=True)`` + ``eval_model``, ``T/data_utils/metrics.py:60-107``): item vectors of every title,
    per-user hits and nDCG, the two means.  fp32 parity mode; north_star tolerance on HR@10: 1e-3."""
    from idvs.morec_amd.data_utils import eval_model, get_item_embeddings
    from idvs.morec_amd.data_utils.metrics import eval_ranks, metrics_from_ranks
    from idvs.morec_amd.model import BertShape, HipBertModel, Model
    from idvs.morec_amd.utils.detgen import det_param
    g = np.load(os.path.join(golden_dir, "g17_eval_modal.npz"))
    S, D, T, item_num, U = (int(v) for v in g["cfg"])
    shape = BertShape.named("micro")
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 num_words_title=T, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                 bert_model_load="bert_micro", word_embedding_dim=shape.hidden_size, compute_dtype="fp32", num_workers=0)
    m = Model(args, item_num, True, HipBertModel(shape), g["pop"])
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(torch.from_numpy(det_param(k, tuple(v.shape))))
    m = m.to(DEV)
    eval_seq = {u: [int(v) for v in g[f"seq.{u}"]] for u in range(U)}
    hist = {u: torch.LongTensor(eval_seq[u][:-1]) for u in range(U)}
    emb = get_item_embeddings(m, g["content"], 16, args, True, DEV)
    err = np.abs(emb.cpu().numpy()[1:] - g["item_embeddings"][1:]).max()      # row 0: the all-[PAD] item (implementation-defined, masked everywhere)
    print(f"g17: item vectors max abs err {err:.2e} (scale {np.abs(g['item_embeddings'][1:]).max():.2e}); min score margin {g['margins'].min():.2e}")
    assert err < 5e-6
    ranks = eval_ranks(m, hist, eval_seq, emb, list(range(U)), args, DEV)
    hit, ndcg = metrics_from_ranks(ranks)
    safe = g["margins"] > 20 * err            # a target whose score sits within fp32 noise of a competitor may legitimately swap with it
    assert safe.sum() >= U - 2
    assert np.array_equal(hit.cpu().numpy()[safe], g["hit_per_user"][safe])
    assert np.abs(ndcg.cpu().numpy()[safe] - g["ndcg_per_user"][safe]).max() < 1e-6
    hit10 = eval_model(m, hist, eval_seq, emb, 16, args, item_num, logging.getLogger("t"), "valid", DEV)
    assert abs(hit10 - float(g["hit10"])) < 1e-3          # north_star: HR@10 within 1e-3
    if safe.all():
        assert abs(hit10 - float(g["hit10"])) < 1e-6


def test_encode_all_items_second_pass_allocates_nothing():
    """``get_item_embeddings`` (``T/data_utils/metrics.py:60-74``) over a catalogue of ragged titles in 16-bit: the chunks are encoded
    largest first, so (a) the second pass of a process asks the device allocator for NOTHING (the driver of round 5 recorded 2.9 s for a
    pass whose kernels take 0.28 s: hipMalloc per chunk on a cold allocator) and (b) an item's vector is the same whatever chunk
    size -- i.e. whatever neighbours and encoding order -- it is computed with (to 16-bit rounding)."""
    from idvs.morec_amd.data_utils import get_item_embeddings
    from idvs.morec_amd.model import BertShape, HipBertModel, Model
    T, item_num, D = 30, 1500, 64
    shape = BertShape.named("micro")
    args = types.SimpleNamespace(max_seq_len=8, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                                 num_words_title=T, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                 bert_model_load="bert_micro", word_embedding_dim=shape.hidden_size, compute_dtype="fp16", num_workers=0)
    torch.manual_seed(5)
    m = Model(args, item_num, True, HipBertModel(shape), np.full(item_num + 1, 1.0 / item_num)).to(DEV)
    rng = np.random.default_rng(3)
    lens = rng.integers(3, T + 1, size=item_num + 1)
    lens[0] = 0
    lens[1000:1256] = T          # one chunk far heavier than the first ones
    ids = rng.integers(5, shape.vocab_size, size=(item_num + 1, T))
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    content = np.concatenate([ids * mask, mask], 1)
    emb = get_item_embeddings(m, content, 256, args, True, DEV)
    torch.cuda.synchronize()
    n0 = torch.cuda.memory_stats(DEV)["num_device_alloc"]
    emb2 = get_item_embeddings(m, content, 256, args, True, DEV)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(DEV)["num_device_alloc"] == n0, "the second encode pass went to hipMalloc"
    assert torch.equal(emb, emb2)
    emb3 = get_item_embeddings(m, content, 100, args, True, DEV)      # other chunks, other order
    # (row 0: the all-[PAD] item, masked everywhere; another chunk size may select another GEMM kernel, i.e. another fp32 summation order:
    # equal to 16-bit rounding, and any misplaced row would be off by O(1))
    assert float((emb[1:] - emb3[1:]).abs().max()) < 1e-2 * float(emb[1:].abs().max())
    assert emb.shape == (item_num + 1, D) and emb.dtype == torch.float32 and bool(torch.isfinite(emb[1:]).all())


@pytest.mark.parametrize("fused", [True, False])
def test_run_driver_id_tower_learns(fused):
    """A few dozen steps of the driver on synthetic data: the loss must fall (both optimisation paths)."""
    from idvs.morec_amd import run
    from idvs.morec_amd.parameters import parse_args
    argv = ["--synthetic", "1500", "--synthetic_items", "300", "--item_tower", "id", "--batch_size", "64", "--embedding_dim",
            "64", "--lr", "3e-3", "--l2_weight", "0.0", "--drop_rate", "0.1", "--epoch", "3", "--compute_dtype", "fp32",
            "--local_rank", "0"] + (["--fused_step"] if fused else [])
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    run.setup_seed(12345)
    best = run.train(args, False, 0)
    assert 0.0 <= best <= 1.0


def test_run_driver_modal_tiny_steps():
    from idvs.morec_amd import run
    from idvs.morec_amd.parameters import parse_args
    args = parse_args(["--synthetic", "400", "--synthetic_items", "200", "--item_tower", "modal", "--bert_model_load", "bert_tiny",
                       "--freeze_paras_before", "0", "--batch_size", "32", "--embedding_dim", "128", "--lr", "1e-3",
                       "--fine_tune_lr", "1e-4", "--epoch", "1", "--max_steps", "6", "--fused_step", "--local_rank", "0"])
    run.setup_seed(12345)
    best = run.train(args, True, 0)
    assert 0.0 <= best <= 1.0


@pytest.mark.parametrize("fused", [True, False])
def test_run_driver_vision_micro_steps(fused):
    """The driver in vision mode (V/run.py shape): uint8 image catalogue, on-device normalisation, Swin micro tower, the
    optimizer grouping of V/run.py:121-135, eval over the image catalogue."""
    from idvs.morec_amd import run
    from idvs.morec_amd.parameters import parse_args
    args = parse_args(["--synthetic", "300", "--synthetic_items", "120", "--item_tower", "modal", "--CV_model_load", "swin_micro",
                       "--CV_resize", "56", "--freeze_paras_before", "0", "--batch_size", "16", "--max_seq_len", "6",
                       "--embedding_dim", "64", "--lr", "1e-3", "--fine_tune_lr", "1e-4", "--epoch", "1", "--max_steps", "4",
                       "--local_rank", "0"] + (["--fused_step"] if fused else []))
    run.setup_seed(12345)
    best = run.train(args, True, 0)
    assert 0.0 <= best <= 1.0


def test_run_driver_bce_loss_learns():
    """The driver with the one-negative BCE loss of bce_text/main-end2end (ID tower): runs, evaluates, loss is finite."""
    from idvs.morec_amd import run
    from idvs.morec_amd.parameters import parse_args
    args = parse_args(["--synthetic", "1200", "--synthetic_items", "300", "--item_tower", "id", "--loss", "bce", "--batch_size", "64",
                       "--embedding_dim", "64", "--lr", "3e-3", "--epoch", "2", "--compute_dtype", "fp32", "--local_rank", "0"])
    run.setup_seed(12345)
    best = run.train(args, False, 0)
    assert 0.0 <= best <= 1.0
