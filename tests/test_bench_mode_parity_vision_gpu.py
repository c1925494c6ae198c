"""-m gpu: the BENCHMARKED vision mode (bf16 operands: what ``bench.py --tower swin_tiny`` times and the default line carries as
``vision_swin_tiny``) against the PARITY mode (exact-fp32 MFMA, pinned to the reference golden g13 at 3e-6 on the loss) on the
SAME kernels the bench runs: Swin-T at 224 x 224, S = 10, D = 2048 (``V/train_swin_tiny.py:22-41``, ``V/parameters.py:34-39``),
16 user sequences = 176 images, so that the stage-1 / stage-2 products (552 k / 138 k rows) run on ``gemm8p_kernel`` /
``gemm_tn8p_kernel`` exactly as in the 704-image bench step (the golden tests are 8 images: two-buffer kernels).
Dropout and DropPath off (their RNG streams cannot be matched between two modes either); same synthetic images and weights.

Stated tolerance of the bf16 vision mode (asserted below, measured values printed): step-0 loss 3e-2, gradient norms of both
optimizer groups 5e-2, loss of steps 0-4 within 1.5 % of the loss, the whole 10-step curve within 8 % (measured on MI355X, round 3,
five runs: 1.2e-2 / 1.1e-2 / <= 0.8 % / 1.5-3.5 %).  At the launcher's learning rates the first ten steps of a randomly initialised
Swin-T at 16 users RAISE the loss, 9.20 -> 10.14, in both modes alike, and that regime amplifies rounding: the fp32 curve repeats
to 4e-4 from run to run, the bf16 curve scatters by 0.17 at step 10 BETWEEN ITS OWN RUNS (order of the fp32 atomic sums under bf16
rounding of the operands), which is the size of its distance to the fp32 curve -- hence the wider bound on the late steps and the
tight one on the early steps, where the two modes are comparable."""
import dataclasses
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(dtype, shape, D, S, item_num, pop, state=None):
    from idvs.morec_amd.model import Model
    from idvs.morec_amd.model.swin import HipSwinForImageClassification
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 CV_model_load="swin_tiny", compute_dtype=dtype)
    torch.manual_seed(12345)
    m = Model(args, item_num, True, HipSwinForImageClassification(shape, D), pop)
    if state is not None:
        m.load_state_dict(state)
    return m.to("cuda").train()


def test_bf16_vision_bench_mode_tracks_fp32_parity_mode():
    from idvs.morec_amd.swin_engine import SwinShape
    from idvs.morec_amd.train_step import TrainStep
    B, S, D, item_num, steps = 16, 10, 2048, 600, 10
    shape = dataclasses.replace(SwinShape.named("swin_tiny"), drop_path_rate=0.0)
    rng = np.random.default_rng(4321)
    ids_all = rng.integers(1, item_num + 1, size=(steps, B, S + 1)).astype(np.int64)
    counts = np.bincount(ids_all.reshape(-1), minlength=item_num + 1).astype(np.float64) + 1.0
    pop = counts / counts[1:].sum()
    pop[0] = 1.0
    gen = torch.Generator(device="cuda").manual_seed(4321)
    catalog = torch.randn((item_num + 1, 3, shape.image_size, shape.image_size), device="cuda", generator=gen)
    catalog[0].zero_()
    m32 = _build("fp32", shape, D, S, item_num, pop)
    state = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
    m16 = _build("bf16", shape, D, S, item_num, pop, state)
    kw = dict(lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=False)

    def batch(i):
        ids = torch.from_numpy(ids_all[i]).cuda().view(-1)
        return ids, catalog[ids], torch.ones(B, S, device="cuda")

    curves, gnorms = {}, {}
    for name, model in (("fp32", m32), ("bf16", m16)):
        ts = TrainStep(model, **kw)
        curves[name] = []
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
    print(f"vision bench-mode parity (Swin-T, {B * (S + 1)} images), bf16 vs fp32 mode: step-0 loss {c16[0]:.5f} vs {c32[0]:.5f} (|d| {d0:.2e}); "
          f"gradient-norm rel. diff {['%.2e' % x for x in gn]}; {steps}-step loss curve max |d| {dcurve:.2e}; "
          f"loss {c32[0]:.4f} -> {c32[-1]:.4f} (fp32), {c16[0]:.4f} -> {c16[-1]:.4f} (bf16); per-step |d| {np.round(np.abs(c16 - c32), 4).tolist()}")
    assert np.isfinite(c16).all() and np.isfinite(c32).all()
    assert d0 < 3e-2, d0
    assert max(gn) < 5e-2, gn
    assert float(np.abs(c16 - c32)[:5].max() / c32.min()) < 1.5e-2, dcurve
    assert float(np.abs(c16 - c32).max() / c32.min()) < 8e-2, dcurve


def test_fp16_vision_mode_tracks_fp32_parity_mode():
    """The same comparison for ``--compute_dtype fp16`` (the reference's own GPU arithmetic, V/run.py autocast + GradScaler; what ``bench.py``'s
    vision lines run): IEEE-half storage / MFMA operands with the loss scale kept on the device.  Stated tolerance: step-0 loss 4e-3 (1/8 of the
    bf16 bound: three more mantissa bits), gradient norms 1e-2, steps 0-4 within 1 % of the loss (measured 0.17 %); the late steps of this loss-raising regime
    scatter with the summation order as in the bf16 test."""
    from idvs.morec_amd.swin_engine import SwinShape
    from idvs.morec_amd.train_step import TrainStep
    B, S, D, item_num, steps = 16, 10, 2048, 600, 10
    shape = dataclasses.replace(SwinShape.named("swin_tiny"), drop_path_rate=0.0)
    rng = np.random.default_rng(4321)
    ids_all = rng.integers(1, item_num + 1, size=(steps, B, S + 1)).astype(np.int64)
    counts = np.bincount(ids_all.reshape(-1), minlength=item_num + 1).astype(np.float64) + 1.0
    pop = counts / counts[1:].sum()
    pop[0] = 1.0
    gen = torch.Generator(device="cuda").manual_seed(4321)
    catalog = torch.randn((item_num + 1, 3, shape.image_size, shape.image_size), device="cuda", generator=gen)
    catalog[0].zero_()
    m32 = _build("fp32", shape, D, S, item_num, pop)
    state = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
    m16 = _build("fp16", shape, D, S, item_num, pop, state)
    kw = dict(lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=False)

    def batch(i):
        ids = torch.from_numpy(ids_all[i]).cuda().view(-1)
        return ids, catalog[ids], torch.ones(B, S, device="cuda")

    curves, gnorms = {}, {}
    for name, model in (("fp32", m32), ("fp16", m16)):
        ts = TrainStep(model, **kw) if name == "fp32" else TrainStep(model, loss_scale=4096.0, **kw)
        assert (ts.sp is not None) == (name == "fp16")
        curves[name] = []
        for i in range(steps):
            loss = ts.forward_backward(*batch(i))
            if i == 0:
                sc = float(ts.sp.host().loss_scale) if ts.sp is not None else 1.0
                gnorms[name] = [float(g["arena"].grad.double().norm()) / sc for g in ts.groups]
            ts.reduce_gradients()
            ts.optimizer_step()
            curves[name].append(float(loss))
        if ts.sp is not None:
            h = ts.sp.host()
            print(f"fp16 vision scaler after {steps} steps: scale {h.loss_scale:g}, applied {h.step}, skipped {h.skipped}")
            assert h.skipped == 0, "a skipped step shifts the fp16 trajectory by one update: lower the starting scale of this test"
        del ts
        torch.cuda.empty_cache()
    c32, c16 = np.array(curves["fp32"]), np.array(curves["fp16"])
    d0 = abs(c16[0] - c32[0])
    gn = [abs(a - b) / b for a, b in zip(gnorms["fp16"], gnorms["fp32"])]
    print(f"vision fp16 vs fp32 mode (Swin-T, {B * (S + 1)} images): step-0 loss {c16[0]:.5f} vs {c32[0]:.5f} (|d| {d0:.2e}); gradient-norm rel. diff "
          f"{['%.2e' % x for x in gn]}; per-step |d| {np.round(np.abs(c16 - c32), 4).tolist()}")
    assert np.isfinite(c16).all()
    assert d0 < 4e-3, d0
    assert max(gn) < 1e-2, gn
    assert float(np.abs(c16 - c32)[:5].max() / c32.min()) < 1e-2      # measured 1.7e-3; the early steps of this loss-raising regime scatter run to run (bf16 test: docstring)
    assert float(np.abs(c16 - c32).max() / c32.min()) < 8e-2
