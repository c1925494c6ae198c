"""-m gpu: ``MOREC_DETERMINISTIC`` (``ops.set_deterministic``).  The reference sets torch's deterministic flags (``T/run.py:313-314``); the
library's counterpart replaces every fp32 atomic of the text / ID / Swin step -- LayerNorm dgamma / dbeta / bias column sums, bias gradients,
position / type rows, the word-table and id-table scatters, multi-block folds of partial rows -- by per-block partials folded in a fixed
order (or a single writer per table row).  Checked: two runs of the same steps from the same initial state are BIT-identical (losses,
every parameter, both AdamW moments), with dropout on and with the weight gradients on the second stream; and the mode changes the
numbers by rounding only (same steps with the mode off: a parameter distance at the level of the atomics' own run-to-run scatter)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pop(ids_all, item_num):
    counts = np.bincount(ids_all.reshape(-1), minlength=item_num + 1).astype(np.float64) + 1.0
    pop = counts / counts[1:].sum()
    pop[0] = 1.0
    return pop


def _run(tower, dtype, steps, det):
    import bench
    from idvs.morec_amd import engine, ops
    from idvs.morec_amd.model import BertShape, HipBertModel, Model
    from idvs.morec_amd.train_step import TrainStep
    B, S, T, D, item_num = 24, 12, 30, 128, 1500
    ids_all = bench.synth_batches(steps, B, S, item_num, np.random.default_rng(2))
    # ragged histories: some users are left-padded (padding item 0), so masked rows / columns and the pad row of the tables take part
    for i in range(steps):
        ids_all[i][::3, :4] = 0
    pop = _pop(ids_all, item_num)
    ops.set_deterministic(det)
    try:
        torch.manual_seed(7)
        if tower == "text":
            content = bench.synth_catalog(item_num, T, np.random.default_rng(1))
            shape = BertShape.named("tiny")
            args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                                         num_words_title=T, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                         bert_model_load="bert_tiny", word_embedding_dim=shape.hidden_size, compute_dtype=dtype)
            m = Model(args, item_num, True, HipBertModel(shape, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1), pop).to(DEV).train()
        elif tower == "vision":
            from idvs.morec_amd.model.swin import HipSwinForImageClassification
            from idvs.morec_amd.swin_engine import SwinShape
            shape = SwinShape.named("swin_micro")      # 56 x 56 images, two stages (plain + shifted windows, a patch merging), DropPath 0.1
            gen = torch.Generator(device=DEV).manual_seed(11)
            catalog = torch.randn((item_num + 1, 3, shape.image_size, shape.image_size), device=DEV, generator=gen)
            catalog[0].zero_()
            args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                                         CV_model_load="swin_micro", compute_dtype=dtype)
            m = Model(args, item_num, True, HipSwinForImageClassification(shape, D), pop).to(DEV).train()
        else:
            args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2, compute_dtype=dtype)
            m = Model(args, item_num, False, None, pop).to(DEV).train()
        ts = TrainStep(m, lr=3e-3, fine_tune_lr=1e-3, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=False,
                       loss_scale=1024.0 if dtype == "fp16" else None)
        losses = []
        for i in range(steps):
            ids = torch.from_numpy(ids_all[i]).to(DEV)
            lm = (ids[:, :-1] != 0).float()
            if tower == "text":
                rows = content[ids_all[i].reshape(-1)]
                pack = tuple(t.to(DEV) for t in engine.token_packing_host(rows[:, T:], rows[:, :T]))
                losses.append(ts.step(ids.view(-1), torch.from_numpy(rows).to(DEV), lm, token_packing=pack))
            elif tower == "vision":
                losses.append(ts.step(ids.view(-1), catalog[ids.view(-1)], lm))
            else:
                losses.append(ts.step(ids.view(-1), ids.view(-1).clone(), lm))
        ts.flush()
        torch.cuda.synchronize()
        state = [t.clone() for g in ts.groups for t in (g["arena"].data, g["arena"].exp_avg, g["arena"].exp_avg_sq)]
        return [float(x) for x in losses], state
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize("tower,dtype", [("text", "fp16"), ("text", "fp32"), ("id", "bf16"), ("vision", "fp16"), ("vision", "fp32")])
def test_two_runs_are_bit_identical(tower, dtype):
    steps = 5
    a = _run(tower, dtype, steps, True)
    b = _run(tower, dtype, steps, True)
    assert a[0] == b[0], (a[0], b[0])
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y), float((x.double() - y.double()).abs().max())
    assert all(np.isfinite(v) for v in a[0]) and a[0][-1] < a[0][0]
    # the mode changes summation orders, nothing else: against the default (atomic) kernels the trajectory agrees to rounding
    c = _run(tower, dtype, steps, False)
    dl = max(abs(u - v) for u, v in zip(a[0], c[0]))
    dp = max(float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)) for x, y in zip(a[1][::3], c[1][::3]))
    print(f"{tower} {dtype}: deterministic x 2 bit-identical over {steps} steps; vs default kernels: max |d loss| {dl:.2e}, parameter distance {dp:.2e}")
    # (measured: text fp16 2.2e-4 / 5.5e-4, text fp32 2.4e-6 / 7.2e-5, id bf16 8.5e-5 / 5.4e-4; a scatter that dropped every source row past
    # the first 32 candidates -- lanes without columns were missing from a ballot -- showed up here as 1e-2 on the id tower)
    # (vision, round 6: the window-attention dbias tiles per workgroup + an ordered fold, the relative-position table as a gather)
    # (vision fp32, later in round 6: the DEFAULT-mode run is not one trajectory but a few discrete ones -- scripts/det_probe.py / det_probe2.py,
    # profiles/r06_det_probe.txt: six runs of the same steps give losses[2] = 5.9386511 or 5.9386406 (and once 5.9386487 with the VALU attention
    # kernels), identical up to there and 4.9e-4 / 8.6e-4 apart (loss / parameters) after five steps.  Where it starts: weight elements of the Swin MLPs
    # whose gradient is ~ 4e-9 ... 1.4e-8, i.e. at AdamW's eps -- a sum of cancelling terms that the atomics' arrival order moves by 0.5 %, which
    # m / (sqrt(v) + eps) turns into parameter differences of 1e-6 ... 5e-6 after ONE step (fp32 rounding would be 1e-9); from there a discrete
    # event picks the branch.  The fp32 attention on the matrix cores (attention_f32mfma.hip) moved the odds between the branches, not their
    # distance.  Hence: the first two losses agree to fp32 rounding, the five-step distance is bounded by the branch distance.)
    if dtype == "fp32":
        assert max(abs(u - v) for u, v in zip(a[0][:2], c[0][:2])) < 5e-6
    fp32_bound = (1.5e-3, 2e-3) if tower == "vision" else (2e-4, 1e-3)
    assert dl < (fp32_bound[0] if dtype == "fp32" else 3e-3) and dp < (fp32_bound[1] if dtype == "fp32" else 4e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_deterministic_kernels_against_fp64(dt):
    """The single-writer / fold-in-order variants themselves against fp64 references, at row widths that leave lanes idle (D = 64: a
    quarter of the wave holds columns) and with heavy index collisions; each call twice: same bits."""
    from idvs.morec_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    ops.set_deterministic(True)
    try:
        for D, R, V in ((64, 500, 40), (128, 2688, 300), (512, 2688, 3000), (520, 77, 9)):
            idx = torch.randint(0, V, (R,), generator=g).to(torch.int32)
            idx[::7] = 5                              # a hot id
            idx[::11] = 0                             # padding id: skipped
            d = (torch.randn(R, D, generator=g)).to(dt).to(DEV)
            ref = torch.zeros(V, D, dtype=torch.float64, device=DEV)
            ref.index_add_(0, idx.long().to(DEV), d.double())
            ref[0] = 0
            outs = []
            for _ in range(2):
                tab = torch.zeros(V, D, device=DEV)
                ops.scatter_add_rows_(d, idx.to(DEV), tab, 0)
                outs.append(tab)
            assert torch.equal(outs[0], outs[1])
            assert float((outs[0].double() - ref).abs().max()) < 1e-4 * float(ref.abs().max()), (D, R)
        # embedding backward: word rows (runs of equal ids in sorted order), position rows, the type row
        n_seq, T, H, V = 300, 30, 768, 200
        ids = torch.randint(0, V, (n_seq * T,), generator=g).to(torch.int32)
        ids[::5] = 101
        dz = torch.randn(n_seq * T, H, generator=g).to(dt).to(DEV)
        order = torch.argsort(ids, stable=True).to(torch.int32).to(DEV)
        ref = torch.zeros(V, H, dtype=torch.float64, device=DEV)
        ref.index_add_(0, ids.long().to(DEV), dz.double())
        ref[0] = 0
        outs = []
        for _ in range(2):
            dw, dp, dty = torch.zeros(V, H, device=DEV), torch.zeros(T + 2, H, device=DEV), torch.zeros(H, device=DEV)
            ops.bert_embed_bwd_(ids.to(DEV), dz, dw, dp, dty, 0, T, order)
            outs.append((dw, dp, dty))
        assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
        assert float((outs[0][0].double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
        assert float((outs[0][1][:T].double() - dz.double().view(n_seq, T, H).sum(0)).abs().max()) < 1e-3
        assert float((outs[0][2].double() - dz.double().sum(0)).abs().max()) < 2e-3
        # column sums over many row blocks
        x = torch.randn(70000, 520, generator=g).to(dt).to(DEV)
        cs = [torch.full((520,), 0.5, device=DEV) for _ in range(2)]
        for c in cs:
            ops.colsum_(x, c)
        assert torch.equal(cs[0], cs[1])
        assert float((cs[0].double() - 0.5 - x.double().sum(0)).abs().max()) < 2e-2
    finally:
        ops.set_deterministic(False)
