"""Vision tower parity on a real MI355X: the Swin kernels (through the C-ABI) against the CPU oracle
(``oracle/morec_oracle/swin_ref.py``) and the drop-in ``Vit_Encoder`` / vision ``Model`` against the golden vectors
captured from the imported reference (``tests/golden/g11_swin_micro.npz``).
Tolerances: exact-fp32 path 2e-4 relative to max-abs (loss 1e-3 absolute per north_star, in practice < 1e-4);
bf16 path 6e-2 relative in the Frobenius norm."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from idvs.morec_amd import ops  # noqa: E402
from idvs.morec_amd.engine import DropCfg  # noqa: E402
from idvs.morec_amd.model import Model  # noqa: E402
from idvs.morec_amd.model.encoders import Vit_Encoder  # noqa: E402
from idvs.morec_amd.model.swin import HipSwinForImageClassification  # noqa: E402
from idvs.morec_amd.swin_engine import SwinShape  # noqa: E402
from idvs.morec_amd.utils.detgen import det_normal, det_param  # noqa: E402
from morec_oracle import swin_ref  # noqa: E402

DEV = "cuda"
DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
T16 = {"bf16": 1.0, "fp16": 0.125}      # 16-bit bounds below are stated for bf16 (8 mantissa bits); IEEE half has 3 more


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def froerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def load_det(module, prefix=""):
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(torch.from_numpy(det_param(prefix + k, tuple(v.shape))))
    return module


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("shift", [0, 3])
@pytest.mark.parametrize("H,W,heads,n_img", [(14, 14, 2, 3), (7, 7, 4, 5), (28, 14, 1, 2)])
def test_window_attention(dt, shift, H, W, heads, n_img):
    if min(H, W) <= 7 and shift:
        pytest.skip("window == map: HF forces shift 0")
    dtype = DT[dt]
    C = heads * 32
    R = n_img * H * W
    qkv = torch.from_numpy(det_normal(f"swa.qkv{H}{W}{heads}", (R, 3 * C), std=0.7)).to(dtype)
    table = torch.from_numpy(det_normal(f"swa.tab{heads}", (169, heads), std=0.5))
    dctx = torch.from_numpy(det_normal(f"swa.dctx{H}{W}{heads}", (R, C), std=1.0)).to(dtype)
    # oracle on the same (storage-rounded) inputs
    q32 = qkv.float().requires_grad_(True)
    t32 = table.clone().requires_grad_(True)
    x = q32.view(n_img, H, W, 3 * C)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    win = swin_ref._partition(x, 7)
    q, k, v = (win[..., i * C:(i + 1) * C].reshape(-1, 49, heads, 32).transpose(1, 2) for i in range(3))
    bias = t32[swin_ref.rel_position_index(7).view(-1)].view(49, 49, heads).permute(2, 0, 1)[None]
    s = q @ k.transpose(-2, -1) * 32 ** -0.5 + bias
    m = swin_ref.shift_mask(H, W, 7, shift)
    if m is not None:
        s = (s.view(n_img, m.shape[0], heads, 49, 49) + m[None, :, None]).view(-1, heads, 49, 49)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(-1, 49, C)
    o = swin_ref._reverse(o, 7, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    ref = o.reshape(R, C)
    (ref * dctx.float()).sum().backward()

    desc = ops.swin_attn_desc(n_img, H, W, 7, shift, heads, 32, dtype)
    bias_t = ops.swin_bias_expand(table.to(DEV), 7)
    np.testing.assert_array_equal(bias_t.cpu().numpy(), bias[0].detach().permute(0, 2, 1).numpy())
    qd = qkv.to(DEV)
    ctx = ops.swin_attn_fwd(desc, qd, bias_t)
    tol = 2e-5 if dt == "fp32" else 1.5e-2 * T16[dt]
    assert relerr(ctx.float().cpu().numpy(), ref.detach().numpy()) < tol
    dbias_t = torch.zeros_like(bias_t)
    dqkv = ops.swin_attn_bwd(desc, qd, bias_t, ctx, dctx.to(DEV), dbias_t)
    dtab = torch.zeros((169, heads), device=DEV)
    ops.swin_bias_reduce_(dbias_t, dtab, 7)
    tolg = 5e-5 if dt == "fp32" else 3e-2 * T16[dt]
    assert froerr(dqkv.float().cpu().numpy(), q32.grad.numpy()) < tolg
    assert froerr(dtab.cpu().numpy(), t32.grad.numpy()) < tolg
    # the same backward with the fused q|k|v bias gradient: same dqkv, same dbias_t, dbqkv += column sums of dqkv (the MFMA path
    # sums the rows before their bf16 rounding: agreement with the sums of the STORED rows to rounding noise, not bit for bit)
    dbias2, dbq = torch.zeros_like(bias_t), torch.full((3 * C,), 0.5, device=DEV)
    dqkv2 = ops.swin_attn_bwd(desc, qd, bias_t, ctx, dctx.to(DEV), dbias2, dbqkv=dbq)
    assert torch.equal(dqkv2, dqkv)
    assert froerr(dbias2.cpu().numpy(), dbias_t.cpu().numpy()) < 1e-5
    want = 0.5 + dqkv.double().sum(0).cpu().numpy()
    assert froerr(dbq.cpu().numpy(), want) < (1e-5 if dt == "fp32" else 3e-3 * T16[dt])
    assert froerr(dbq.cpu().numpy() - 0.5, q32.grad.double().sum(0).numpy()) < tolg


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_patchify_merge_pool_residual(dt):
    dtype = DT[dt]
    n, R, ps = 3, 56, 4
    px = torch.from_numpy(det_normal("sw.px", (n, 3, R, R)))
    pat = ops.swin_patchify(px.to(DEV), ps, dtype)
    ref = torch.nn.functional.unfold(px, kernel_size=ps, stride=ps).transpose(1, 2).reshape(-1, 3 * ps * ps)   # (c, i, j) column order
    np.testing.assert_array_equal(pat.float().cpu().numpy(), ref.to(dtype).float().numpy())
    H, W, C = 14, 10, 32
    x = torch.from_numpy(det_normal("sw.mx", (n * H * W, C))).to(dtype)
    mg = ops.swin_merge(x.to(DEV), n, H, W, C)
    g = x.view(n, H, W, C)
    refm = torch.cat([g[:, r::2, c::2, :] for c in range(2) for r in range(2)], -1).reshape(-1, 4 * C)
    np.testing.assert_array_equal(mg.float().cpu().numpy(), refm.float().numpy())
    back = ops.swin_merge(mg, n, H, W, C, reverse=True)
    np.testing.assert_array_equal(back.float().cpu().numpy(), x.float().numpy())
    T = 49
    y = torch.from_numpy(det_normal("sw.pool", (n * T, C))).to(dtype)
    pooled = ops.swin_pool_fwd(y.to(DEV), n, T)
    assert relerr(pooled.float().cpu().numpy(), y.float().view(n, T, C).mean(1).numpy()) < (1e-6 if dt == "fp32" else 8e-3 * T16[dt])
    d = torch.from_numpy(det_normal("sw.dpool", (n, C))).to(dtype)
    dx = ops.swin_pool_bwd(d.to(DEV), n, T)
    assert relerr(dx.float().cpu().numpy(), (d.float() / T)[:, None, :].expand(n, T, C).reshape(-1, C).numpy()) < (1e-6 if dt == "fp32" else 8e-3 * T16[dt])
    a = torch.from_numpy(det_normal("sw.a", (n * T, C))).to(dtype)
    bias = torch.from_numpy(det_normal("sw.b", (C,)))
    sc = torch.tensor([0.0, 1.25, 1.25])
    out = ops.bias_residual(a.to(DEV), bias.to(DEV), y.to(DEV), sc.to(DEV), T, inplace=False)
    refo = y.float() + sc.repeat_interleave(T)[:, None] * (a.float() + bias)
    assert relerr(out.float().cpu().numpy(), refo.numpy()) < (1e-6 if dt == "fp32" else 8e-3 * T16[dt])


def test_droppath_scale_statistics():
    s = ops.droppath_scale(200000, 0.1, 1234).cpu().numpy()
    keep_scale = np.float32(1.0) / (np.float32(1.0) - np.float32(int(0.1 * 65536)) / np.float32(65536.0))   # exact for the applied probability
    assert set(np.unique(s).tolist()) <= {0.0, float(keep_scale)}
    assert abs((s == 0).mean() - 0.1) < 0.005
    s2 = ops.droppath_scale(200000, 0.1, 1234).cpu().numpy()
    np.testing.assert_array_equal(s, s2)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_layernorm_rowscale_dres(dt):
    """Pre-LN residual form: z = res + rowscale * (x + bias), y = LN(z); backward with the residual-stream gradient dres."""
    dtype = DT[dt]
    M, N, rps = 12, 64, 4
    x = torch.from_numpy(det_normal("lnr.x", (M, N))).to(dtype)
    res = torch.from_numpy(det_normal("lnr.r", (M, N))).to(dtype)
    bias = torch.from_numpy(det_normal("lnr.b", (N,), std=0.3))
    gam = torch.from_numpy(1 + 0.1 * det_normal("lnr.g", (N,)))
    bet = torch.from_numpy(det_normal("lnr.be", (N,), std=0.1))
    sc = torch.tensor([1.25, 0.0, 1.25])
    dy = torch.from_numpy(det_normal("lnr.dy", (M, N))).to(dtype)
    dres = torch.from_numpy(det_normal("lnr.dres", (M, N))).to(dtype)
    xr, br, gr, ber = x.float().requires_grad_(True), bias.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rr = res.float().requires_grad_(True)
    z = rr + sc.repeat_interleave(rps)[:, None] * (xr + br)
    zs = z + (z.to(dtype).float() - z).detach()          # the kernel normalises what it stored
    y = torch.nn.functional.layer_norm(zs, (N,), gr, ber, 1e-5)
    ((y * dy.float()).sum() + (z * dres.float()).sum()).backward()
    yk, zk, mean, rstd = ops.layernorm_fwd(x.to(DEV), gam.to(DEV), bet.to(DEV), 1e-5, bias=bias.to(DEV), res=res.to(DEV),
                                           rowscale=sc.to(DEV), rows_per_scale=rps)
    tol = 2e-6 if dt == "fp32" else 1e-2
    assert relerr(zk.float().cpu().numpy(), z.detach().numpy()) < tol
    assert relerr(yk.float().cpu().numpy(), y.detach().numpy()) < tol * 2
    dg, db, dbias = (torch.zeros(N, device=DEV) for _ in range(3))
    dz, dzd = ops.layernorm_bwd(dy.to(DEV), None, zk, mean, rstd, gam.to(DEV), dg, db, dbias=dbias, dres=dres.to(DEV),
                                rowscale=sc.to(DEV), rows_per_scale=rps)
    tolg = 1e-5 if dt == "fp32" else 2e-2
    assert froerr(dz.float().cpu().numpy(), rr.grad.numpy()) < tolg
    assert froerr(dzd.float().cpu().numpy(), xr.grad.numpy()) < tolg
    assert froerr(dbias.cpu().numpy(), br.grad.numpy()) < tolg
    assert froerr(dg.cpu().numpy(), gr.grad.numpy()) < tolg
    assert froerr(db.cpu().numpy(), ber.grad.numpy()) < tolg


# ------------------------------------------------------------------------------------------------ encoder / model vs goldens
def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "g11_swin_micro.npz"))


def _shape_of(G, tag):
    R0, ps, ed, ws, N, D = (int(v) for v in G[f"{tag}.cfg"])
    return SwinShape(image_size=R0, patch_size=ps, embed_dim=ed, depths=tuple(int(d) for d in G[f"{tag}.depths"]),
                     num_heads=tuple(int(h) for h in G[f"{tag}.heads"]), window_size=ws), N, D


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["m2", "m3"])
def test_g11_vit_encoder_golden(golden_dir, tag, dt):
    G = _golden(golden_dir)
    shape, N, D = _shape_of(G, tag)
    enc = Vit_Encoder(HipSwinForImageClassification(shape, D), compute_dtype=DT[dt])
    load_det(enc, "cv_encoder.").to(DEV).eval()
    x = torch.from_numpy(det_normal(f"g11{tag}.x", (N, 3, shape.image_size, shape.image_size), std=1.0)).to(DEV)
    R = torch.from_numpy(det_normal(f"g11{tag}.R", (N, D), std=1.0)).to(DEV)
    y = enc(x)
    e_y = relerr(y.detach().cpu().numpy(), G[f"{tag}.y"])
    assert e_y < (1e-4 if dt == "fp32" else 5e-2), e_y
    (y * R).sum().backward()
    worst = 0.0
    for n, p in enc.named_parameters():
        ref = float(G[f"{tag}.grad_norm.cv_encoder.{n}"])
        got = float(p.grad.double().norm())
        worst = max(worst, abs(got - ref) / (ref + 1e-12))
        assert abs(got - ref) <= (2e-3 if dt == "fp32" else 8e-2) * ref + (1e-6 if dt == "fp32" else 1e-3), (n, got, ref)
    for k in G.files:
        if k.startswith(f"{tag}.grad."):
            n = k[len(f"{tag}.grad.cv_encoder."):]
            got = dict(enc.named_parameters())[n].grad.cpu().numpy()
            if np.abs(G[k]).max() < 1e-7:      # key-bias gradients are mathematically zero (softmax shift invariance)
                assert np.abs(got).max() < (1e-6 if dt == "fp32" else 1e-2), n
                continue
            assert froerr(got, G[k]) < (5e-4 if dt == "fp32" else 8e-2), (n, froerr(got, G[k]))
    print(f"g11 {tag} {dt}: y relerr {e_y:.2e}, worst grad-norm err {worst:.2e}")


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_g11_vision_model_loss_golden(golden_dir, dt):
    G = _golden(golden_dir)
    S, D, item_num, B = (int(v) for v in G["full.cfg"])
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 CV_model_load="swin_micro", compute_dtype=dt)
    net = HipSwinForImageClassification(SwinShape.named("swin_micro"), D)
    m = Model(args, item_num, True, net, G["full.pop"].tolist())
    load_det(m).to(DEV).eval()
    ids, log_mask = G["full.ids"], G["full.log_mask"]
    images = det_normal("g11f.images", (item_num + 1, 3, 56, 56), std=1.0).astype(np.float32)
    images[0] = 0.0
    px = torch.from_numpy(images[ids.reshape(-1)]).to(DEV)
    loss = m(torch.from_numpy(ids).view(-1).to(DEV), px, torch.from_numpy(log_mask).to(DEV), DEV)
    assert abs(float(loss.detach()) - float(G["full.loss"])) < (1e-3 if dt == "fp32" else 5e-2), (float(loss), float(G["full.loss"]))
    loss.backward()
    for n, p in m.named_parameters():
        ref = float(G[f"full.grad_norm.{n}"])
        got = float(p.grad.double().norm())
        assert abs(got - ref) <= (3e-3 if dt == "fp32" else 1e-1) * ref + (1e-6 if dt == "fp32" else 1e-3), (n, got, ref)


def test_droppath_training_matches_oracle_with_exported_scales():
    """Training mode: the per-image DropPath scales the kernels draw are re-created through the C-ABI and fed to the
    oracle, which must then reproduce forward and gradients (fp32)."""
    shape = SwinShape(image_size=56, embed_dim=32, depths=(2, 2), num_heads=(1, 2), drop_path_rate=0.5)
    N, D = 6, 48
    enc = Vit_Encoder(HipSwinForImageClassification(shape, D), compute_dtype=torch.float32)
    load_det(enc, "cv_encoder.").to(DEV).train()
    x = torch.from_numpy(det_normal("dp.x", (N, 3, 56, 56)))
    R = torch.from_numpy(det_normal("dp.R", (N, D)))
    drop = DropCfg(0.0, 0.0, 0xABCDEF12345)
    y = enc.encode(x.to(DEV), drop)
    (y.float() * R.to(DEV)).sum().backward()
    rates = shape.drop_path_rates()
    scales = [None if r == 0 else ops.droppath_scale(N, r, drop.site(i)).cpu() for i, r in enumerate(rates)]
    assert any(s is not None and (s == 0).any() for s in scales) and any(s is not None and (s > 0).any() for s in scales)
    p = {"cv_encoder." + n: t.detach().cpu().clone().requires_grad_(True) for n, t in enc.named_parameters()}
    cfg = swin_ref.SwinCfg(image_size=56, embed_dim=32, depths=(2, 2), num_heads=(1, 2))
    ks = [torch.ones(N) if s is None else s for s in scales]
    yr = swin_ref.vit_encoder_forward(p, cfg, x, keep_scales=ks)
    assert relerr(y.float().cpu().detach().numpy(), yr.detach().numpy()) < 1e-4
    (yr * R).sum().backward()
    for n, t in enc.named_parameters():
        ref = p["cv_encoder." + n].grad
        assert froerr(t.grad.cpu().numpy(), ref.numpy()) < 1e-3 or float(ref.norm()) < 1e-7, n


def _full_size_swin_golden(golden_dir, dt, name, fname, tag):
    G = np.load(os.path.join(golden_dir, fname))
    S, D, item_num, B = (int(v) for v in G["cfg"])
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 CV_model_load=name, compute_dtype=dt)
    m = Model(args, item_num, True, HipSwinForImageClassification(SwinShape.named(name), D), G["pop"].tolist())
    load_det(m).to(DEV).eval()
    ids, log_mask = G["ids"], G["log_mask"]
    images = det_normal(f"{tag}.images", (item_num + 1, 3, 224, 224), std=1.0).astype(np.float32)
    images[0] = 0.0
    px = torch.from_numpy(images[ids.reshape(-1)]).to(DEV)
    with torch.no_grad():
        vec = m.cv_encoder(px)
    e_v = relerr(vec[:, :8].cpu().numpy(), G["item_vec_probe"])
    half = dt in ("bf16", "fp16")
    k16 = T16.get(dt, 1.0)
    assert e_v < (2e-4 if not half else 6e-2 * k16), e_v
    loss = m(torch.from_numpy(ids).view(-1).to(DEV), px, torch.from_numpy(log_mask).to(DEV), DEV)
    e_l = abs(float(loss.detach()) - float(G["loss"]))
    assert e_l < (1e-3 if not half else 5e-2 * k16), (float(loss.detach()), float(G["loss"]))   # north_star: loss within 1e-3 in fp32
    gs = 1024.0 if dt == "fp16" else 1.0      # fp16: a fixed loss scale keeps the activation gradients out of the subnormals (the training step's GradScaler does it dynamically)
    (loss * gs).backward()
    worst = 0.0
    for n, p in m.named_parameters():
        ref = float(G[f"grad_norm.{n}"])
        got = float(p.grad.double().norm()) / gs
        worst = max(worst, abs(got - ref) / (ref + 1e-9)) if ref > 1e-6 else worst
        kg = 1.0 if dt == "bf16" else 0.25      # fp16 gradient norms: measured worst 2.4e-2 (g13) / 2.7e-3 (g15); a quarter of the bf16 bound
        assert abs(got - ref) <= (5e-3 if not half else 1.5e-1 * kg) * ref + (1e-6 if not half else 1e-3), (n, got, ref)
    floor = ""
    if half:
        # yardstick g21 (tests/golden/make_autocast_floor_vision.py): the reference's OWN loss under torch.autocast(fp16 / bf16) on these inputs --
        # how far 16-bit GEMM operands move the reference itself from its fp32 loss.  The HIP 16-bit modes must sit inside 1.5 x that gap.
        import json
        with open(os.path.join(golden_dir, "g21_autocast_floor_vision.json")) as fh:
            fl = json.load(fh)[name]
        ref_gap = abs(fl["autocast_" + dt] - fl["fp32"])
        floor = f", the reference's own autocast gap {ref_gap:.2e}"
        assert abs(fl["fp32"] - float(G["loss"])) < 1e-5
        assert e_l <= 1.5 * ref_gap + 1e-4, (e_l, ref_gap)
    print(f"{tag} {name} {dt}: item-vector relerr {e_v:.2e}, |loss - ref| {e_l:.2e}, worst grad-norm relerr {worst:.2e}{floor}")


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16", "fp32x3"])      # fp32x3: the fp32 bounds (GEMMs as three bf16 MFMA passes over operand splits); fp16: 1/8 of the bf16 bounds
def test_g13_swin_tiny_full_size_golden(golden_dir, dt):
    """Full-size Swin-T tower (real config, 224 x 224) in the vision Model against the reference's scalars."""
    _full_size_swin_golden(golden_dir, dt, "swin_tiny", "g13_swin_tiny_scalars.npz", "g13")


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_g15_swin_base_full_size_golden(golden_dir, dt):
    """BASELINE.json configs[4]: full-size Swin-B tower (pretrained_models/swin_base/config.json: embed 128, depths 2/2/18/2,
    heads 4/8/16/32; V/run.py:47-54) in the vision Model against scalars captured from the reference + installed HF Swin
    (tests/golden/make_golden_vision.py --only g15)."""
    _full_size_swin_golden(golden_dir, dt, "swin_base", "g15_swin_base_scalars.npz", "g15")


def test_patchify_from_uint8_images_is_bit_exact():
    """Decoded uint8 HWC images -> patch rows with ToTensor + Normalize(0.5, 0.5) fused (V/data_utils/dataset.py:69-73): the
    fp32 rows equal patchifying the host-normalised NCHW tensor bit for bit; the encoder accepts either input."""
    n, R = 3, 56
    u8 = torch.from_numpy((np.abs(det_normal("u8.img", (n, R, R, 3))) * 97).astype(np.int64) % 256).to(torch.uint8)
    host = ((u8.permute(0, 3, 1, 2).float() / 255.0) - 0.5) / 0.5
    a = ops.swin_patchify_u8(u8.to(DEV), 4, torch.float32)
    b = ops.swin_patchify(host.contiguous().to(DEV), 4, torch.float32)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    shape = SwinShape.named("swin_micro")
    enc = Vit_Encoder(HipSwinForImageClassification(shape, 32), compute_dtype=torch.float32)
    load_det(enc, "cv_encoder.").to(DEV).eval()
    with torch.no_grad():
        np.testing.assert_array_equal(enc(u8.to(DEV)).cpu().numpy(), enc(host.contiguous().to(DEV)).cpu().numpy())
