"""CPU: host logic added in round 2 -- the reference's sampler semantics and the frozen-prefix bookkeeping."""
import torch
from torch.utils.data.distributed import DistributedSampler

from idvs.morec_amd.data_utils import epoch_batches
from idvs.morec_amd.engine import TE, bert_grad_from, bert_needs_grad_buffer


def test_epoch_batches_are_the_reference_sampler_and_loader():
    """T/run.py:114,123-124,230: DistributedSampler(dataset) (shuffle, seed 0 + epoch, padded to a multiple of the world size) cut
    by DataLoader(batch_size, no drop_last): every rank sees ceil(n / world) samples per epoch and a SHORT last batch."""
    n, B, world = 103, 8, 4
    for epoch in (1, 2, 7):
        seen = []
        for rank in range(world):
            batches = epoch_batches(n, B, world, rank, epoch)
            s = DistributedSampler(range(n), num_replicas=world, rank=rank)     # what the reference constructs
            s.set_epoch(epoch)
            want = list(iter(s))
            dl = torch.utils.data.DataLoader(range(n), batch_size=B, sampler=s)
            assert [list(map(int, b)) for b in dl] == batches
            assert sum(len(b) for b in batches) == len(want) == 26 and len(batches[-1]) == 26 % B
            seen += [i for b in batches for i in b]
        assert len(seen) == 104 and set(seen) == set(range(n))                 # padded by repeating one index
    assert epoch_batches(n, B, world, 0, 1) != epoch_batches(n, B, world, 0, 2)    # set_epoch reshuffles


def test_bert_grad_from_follows_the_freeze_index():
    bm = TE + "bert_model."
    emb = [bm + "embeddings.word_embeddings.weight", bm + "embeddings.LayerNorm.bias"]
    lay = lambda l: [bm + f"encoder.layer.{l}.attention.self.query.weight", bm + f"encoder.layer.{l}.output.LayerNorm.bias"]
    head = [TE + "fc.weight", "user_encoder.transformer_encoder.layer_norm.weight"]
    assert bert_grad_from(emb + lay(0) + lay(11) + head, 12) == -1          # --freeze_paras_before 0 (the launcher's value)
    assert bert_grad_from(lay(10) + lay(11) + head, 12) == 10               # 165 = 5 + 16 * 10 (the parser's default)
    assert bert_grad_from(head + [bm + "pooler.dense.weight"], 12) == 12    # nothing inside bert_model trains
    assert not bert_needs_grad_buffer(emb[0], 10) and not bert_needs_grad_buffer(lay(9)[0], 10)
    assert bert_needs_grad_buffer(lay(10)[0], 10) and bert_needs_grad_buffer(head[0], 10) and bert_needs_grad_buffer(emb[0], -1)
    assert not bert_needs_grad_buffer(bm + "pooler.dense.bias", -1)


def test_pretrained_bert_key_normalisation_and_checkpoint_dir():
    """a stock bert-base-uncased file names its LayerNorm parameters ``*.LayerNorm.gamma`` / ``.beta`` under a ``bert.``
    prefix (HF renames them at load time, ``T/run.py:51-53``); the tower must load ALL of them, and a file that misses a parameter is
    refused instead of silently keeping the random initialisation.  The checkpoint directory is the reference's (``T/run.py:326-337``)."""
    import types

    import pytest
    import torch

    from idvs.morec_amd import run
    from idvs.morec_amd.model import BertShape, HipBertModel
    shape = BertShape(vocab_size=50, hidden_size=16, num_hidden_layers=2, num_attention_heads=2, intermediate_size=32, max_position_embeddings=12)
    src, dst = HipBertModel(shape), HipBertModel(shape)
    with torch.no_grad():
        for p in src.parameters():
            p.uniform_(-1, 1)
    stock = {}
    for k, v in src.state_dict().items():
        k = k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
        stock["bert." + k] = v.clone()
    stock["cls.predictions.bias"] = torch.zeros(50)
    stock["bert.embeddings.position_ids"] = torch.arange(12)[None]
    sd = run.normalize_pretrained_bert_keys(stock)
    assert set(sd) == set(src.state_dict())
    missing, unexpected = dst.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # a file without the LayerNorm parameters is refused by the driver's loader
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "bert_x"))
        torch.save({k: v for k, v in stock.items() if "LayerNorm" not in k}, os.path.join(d, "bert_x", "pytorch_model.bin"))
        args = types.SimpleNamespace(pretrained_dir=d, bert_model_load="bert_x")
        with pytest.raises(SystemExit):
            run._load_pretrained_text_tower(HipBertModel(shape), args)
        torch.save(stock, os.path.join(d, "bert_x", "pytorch_model.bin"))
        assert run._load_pretrained_text_tower(HipBertModel(shape), args) is True
    a = types.SimpleNamespace(item_tower="modal", bert_model_load="bert_base_uncased", CV_model_load="None", freeze_paras_before=165,
                              embedding_dim=512, batch_size=128, lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.02,
                              checkpoint_root=".")
    assert run.model_dir_of(a, 8) == "./checkpoint_modal_bert_base_uncased_freeze_165/cpt_bert_base_uncased_ed_512_bs_1024_lr_0.0001_Flr_5e-05_L2_0.01_FL2_0.02"
    a.item_tower = "id"
    assert run.model_dir_of(a, 1) == "./checkpoint_id/cpt_id_ed_512_bs_128_lr_0.0001_Flr_5e-05_L2_0.01_FL2_0.02"
    # the vision run's label (V/run.py:307-324: '.pth' stripped, '<model>-<freeze>' prefix) -- pinned string for string; an older
    # spelling of this driver and V's ID label (no GPU factor) are searched when LOADING
    v = types.SimpleNamespace(item_tower="modal", bert_model_load="bert_base_uncased", CV_model_load="swin_tiny.pth", freeze_paras_before=0,
                              embedding_dim=2048, batch_size=64, lr=1e-4, fine_tune_lr=1e-4, l2_weight=0.1, fine_tune_l2_weight=0.0,
                              checkpoint_root=".")
    cands = run.model_dir_candidates(v, 4)
    assert cands[0] == "./checkpoint_modal_swin_tiny_freeze_0/cpt_swin_tiny-0_ed_2048_bs_256_lr_0.0001_Flr_0.0001_L2_0.1_FL2_0.0"
    assert run.model_dir_of(v, 4) == cands[0]
    assert "./checkpoint_modal_swin_tiny.pth_freeze_0/cpt_swin_tiny.pth_ed_2048_bs_256_lr_0.0001_Flr_0.0001_L2_0.1_FL2_0.0" in cands[1:]
    a.item_tower = "id"
    assert run.model_dir_candidates(a, 4) == ["./checkpoint_id/cpt_id_ed_512_bs_512_lr_0.0001_Flr_5e-05_L2_0.01_FL2_0.02",
                                              "./checkpoint_id/cpt_id_ed_512_bs_128_lr_0.0001_Flr_5e-05_L2_0.01_FL2_0.02"]


def test_token_packing_host_equals_device_bookkeeping():
    """The collate-side (numpy) unpadded-layout bookkeeping = the device-side one (`engine.token_packing`): same row offsets and packed-row
    indices for ragged titles, the all-[PAD] padding item (keeps its first position) and full-length rows; a mask with holes is refused."""
    import numpy as np
    import torch

    from idvs.morec_amd import engine
    rng = np.random.default_rng(3)
    Nc, T = 57, 30
    lens = rng.integers(0, T + 1, Nc)
    lens[0], lens[5], lens[9] = 0, T, 1
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    cu_h, tok_h = engine.token_packing_host(mask)
    cu_d, tok_d = engine.token_packing(torch.from_numpy(mask))
    assert cu_h.dtype == torch.int32 and tok_h.dtype == torch.int32
    assert torch.equal(cu_h, cu_d) and torch.equal(tok_h, tok_d)
    assert int(cu_h[-1]) == int(np.maximum(lens, 1).sum())
    ids = rng.integers(0, 40, (Nc, T))
    cu_o, tok_o, order, inv = engine.token_packing_host(mask, ids)
    assert torch.equal(cu_o, cu_h) and torch.equal(tok_o, tok_h) and order.dtype == torch.int32 and inv.dtype == torch.int32
    assert torch.equal(inv[tok_h.long()], torch.arange(tok_h.numel(), dtype=torch.int32))       # padded row -> packed row ...
    assert int((inv >= 0).sum()) == tok_h.numel() and int(inv.min()) == -1                     # ... and -1 on every [PAD] row
    flat = ids.reshape(-1)
    assert sorted(order.tolist()) == list(range(Nc * T)) and (np.diff(flat[order.numpy()]) >= 0).all()      # a permutation that groups equal ids
    holes = mask.copy()
    holes[3, 1] = 0
    holes[3, 4] = 1
    assert engine.token_packing_host(holes) is None


def test_fp32_gemm_mode_is_scoped_and_resolved_from_compute_dtype():
    """``compute_dtype`` "fp32x3" = fp32 tensors + the bf16x3 GEMM mode; the mode is set for a block and restored (nesting included)."""
    import types

    import torch

    from idvs.morec_amd import ops
    from idvs.morec_amd.model.encoders import resolve_dtype, resolve_fp32_gemm
    for name, dt, mode in (("bf16", torch.bfloat16, "exact"), ("fp32", torch.float32, "exact"), ("fp32x3", torch.float32, "bf16x3")):
        a = types.SimpleNamespace(compute_dtype=name)
        assert resolve_dtype(a) == dt and resolve_fp32_gemm(a) == mode
    assert ops.FP32_GEMM == "exact"
    with ops.fp32_gemm_mode("bf16x3"):
        assert ops.FP32_GEMM == "bf16x3"
        with ops.fp32_gemm_mode("exact"):
            assert ops.FP32_GEMM == "exact"
        assert ops.FP32_GEMM == "bf16x3"
    assert ops.FP32_GEMM == "exact"
    try:
        with ops.fp32_gemm_mode("bf16x3"):
            raise KeyError("x")
    except KeyError:
        pass
    assert ops.FP32_GEMM == "exact"


def test_weight_gradient_split_fills_the_chip_once():
    """``engine._splitk``: token chunks x 256 x 256 tiles of dW = dY^T X never exceed the 256 CUs in the large-tile regime (260 workgroups
    -- 10 tiles x 26 chunks, the [1152, 384] gradient of Swin-T stage 3 -- ran as two rounds), and BERT-base keeps its measured splits."""
    from idvs.morec_amd.engine import _cdiv, _splitk
    for N, K, M, want in [(768, 768, 55000, 28), (2304, 768, 55000, 9), (3072, 768, 55000, 7), (768, 3072, 55000, 7)]:
        assert _splitk(N, K, M) == want
    shapes = [(c * a, c * b, rows) for c, rows in ((96, 2207744), (192, 551936), (384, 137984), (768, 34496), (128, 1103872), (256, 275968),
                                                   (512, 68992), (1024, 17248)) for a, b in ((3, 1), (1, 1), (4, 1), (1, 4))]
    for N, K, M in shapes:
        s = _splitk(N, K, M)
        big = _cdiv(N, 256) * _cdiv(K, 256)
        assert s >= 1
        if big * s >= 192:                      # the 256 x 256-tile kernel's regime
            assert big * s <= 256, (N, K, M, s)


def test_run_defaults_to_the_reference_arithmetic():
    """``run.py --compute_dtype`` defaults to fp16: the reference's autocast + GradScaler step (T/run.py:210,242-247), the mode bench.py times."""
    from idvs.morec_amd.parameters import parse_args
    a = parse_args([])
    assert a.compute_dtype == "fp16" and a.collate_workers == 2
