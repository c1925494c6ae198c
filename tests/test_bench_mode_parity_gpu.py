"""-m gpu: the BENCHMARKED mode (bf16 operands, what bench.py times) against the PARITY mode (exact-fp32 MFMA, the one pinned to
the reference goldens at 1e-4 on the loss) AT THE BENCH CONFIGURATION -- BERT-base, B = 128 user sequences, S = 20, T = 30,
D = 512 (BASELINE.json configs[2]; T/train_bert_base.py:22-28) -- on the same synthetic MIND-shaped batches, dropout off:
step-0 loss, gradient norms of both optimizer groups, and a 20-step loss curve under the fused AdamW step.
The bounds asserted are the measured deviations of the bf16 mode with ~2x headroom (printed by the test), i.e. the stated
tolerance of the number bench.py reports: step-0 loss 3e-2, gradient norms 5e-2, 20-step loss curve 1.5e-1 absolute / 2 % relative
(the step-0 deviation is rounding-pattern dependent: 1.3e-3 ... 1.5e-2 measured across GEMM summation orders)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _build(dtype, shape, item_num, pop, S, T, D, state=None):
    from idvs.morec_amd.model import HipBertModel, Model
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 num_words_title=T, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                 bert_model_load="bert_base", word_embedding_dim=shape.hidden_size, compute_dtype=dtype)
    torch.manual_seed(12345)
    m = Model(args, item_num, True, HipBertModel(shape, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), pop)
    if state is not None:
        m.load_state_dict(state)
    return m.to("cuda").train()


def test_bf16_bench_mode_tracks_fp32_parity_mode_at_bench_config():
    import bench
    from idvs.morec_amd.model import BertShape
    from idvs.morec_amd.train_step import TrainStep
    B, S, T, D, item_num, steps = 128, 20, 30, 512, 20000, 20
    shape = BertShape.named("base")
    rng = np.random.default_rng(12345)
    content = bench.synth_catalog(item_num, T, rng)
    ids_all = bench.synth_batches(steps, B, S, item_num, np.random.default_rng(13345))
    counts = np.bincount(ids_all.reshape(-1), minlength=item_num + 1).astype(np.float64) + 1.0
    pop = counts / counts[1:].sum()
    pop[0] = 1.0
    m32 = _build("fp32", shape, item_num, pop, S, T, D)
    state = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
    m16 = _build("bf16", shape, item_num, pop, S, T, D, state)
    kw = dict(lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=False)
    ts32, ts16 = TrainStep(m32, **kw), TrainStep(m16, **kw)

    def batch(i):
        ids = torch.from_numpy(ids_all[i]).cuda()
        items = torch.from_numpy(content[ids_all[i].reshape(-1)]).cuda()
        return ids.view(-1), items, torch.ones(B, S, device="cuda")

    curves, gnorms = {"fp32": [], "bf16": []}, {}
    for name, ts in (("fp32", ts32), ("bf16", ts16)):
        for i in range(steps):
            loss = ts.forward_backward(*batch(i))
            if i == 0:
                gnorms[name] = [float(g["arena"].grad.double().norm()) for g in ts.groups]
            ts.reduce_gradients()
            ts.optimizer_step()
            curves[name].append(float(loss))
        del ts
    torch.cuda.empty_cache()
    c32, c16 = np.array(curves["fp32"]), np.array(curves["bf16"])
    d0 = abs(c16[0] - c32[0])
    dcurve = float(np.abs(c16 - c32).max())
    gn = [abs(a - b) / b for a, b in zip(gnorms["bf16"], gnorms["fp32"])]
    print(f"bench-config parity, bf16 vs fp32 mode: step-0 loss {c16[0]:.5f} vs {c32[0]:.5f} (|d| {d0:.2e}); gradient-norm rel. diff "
          f"tower {gn[0]:.2e}, recommender {gn[1]:.2e}; 20-step loss curve max |d| {dcurve:.2e}; "
          f"loss {c32[0]:.4f} -> {c32[-1]:.4f} (fp32), {c16[0]:.4f} -> {c16[-1]:.4f} (bf16)")
    assert np.isfinite(c16).all() and np.isfinite(c32).all()
    assert c32[-1] < c32[0] - 0.05, "the fp32 parity mode does not train on these batches"
    # Measured on MI355X (round 2).  The step-0 loss of the bf16 mode depends on WHICH rounding pattern the GEMM kernels produce: with
    # one fp32 summation order it sits 1.3e-3 from the fp32 mode, with another (the K-split of the tail round: the same error against
    # an exact product, 0.02 % of the elements rounded the other way) 1.5e-2 -- scripts/split_divergence.py: 9.9532 / 9.9627 /
    # 9.9568 / 9.9546 for four split points.  That spread (0.15 % of the loss) is the rounding-noise floor of a 12-layer bf16
    # encoder at random init, so the stated tolerance is 3e-2 on the step-0 loss, 5e-2 on the gradient norms (measured 0.6e-2 ...
    # 2.3e-2), and the 20-step curve within 1.5e-1 / 2 % of the loss (measured 6e-2 ... 7.4e-2 while the loss falls 10.90 -> 7.90).
    assert d0 < 3e-2, d0
    assert max(gn) < 5e-2, gn
    assert dcurve < 1.5e-1, dcurve
    assert float(np.abs(c16 - c32).max() / c32.min()) < 2e-2
