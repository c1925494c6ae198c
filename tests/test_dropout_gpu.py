"""Training-mode dropout in the HIP path.  The reference's RNG streams cannot be reproduced, so parity is checked
(a) statistically for the counter-based mask generator, (b) exactly against PyTorch references that are handed the
SAME masks (exported by the library's test hook), and (c) end to end by a directional-derivative check of the
whole model under a frozen dropout stream."""
import math
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

from idvs.morec_amd import ops  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_mask_statistics():
    n = 1 << 22
    for p in (0.1, 0.5):
        m = ops.dropout_keep_mask(n, p, 1234567).float()
        assert abs(m.mean().item() - (1 - p)) < 2e-3
        assert abs(((m[1:] - m.mean()) * (m[:-1] - m.mean())).mean().item()) < 2e-3          # lag-1 correlation
        assert abs(((m[768:] - m.mean()) * (m[:-768] - m.mean())).mean().item()) < 2e-3      # row-to-row correlation
        m2 = ops.dropout_keep_mask(n, p, 1234568).float()
        assert abs(((m - m.mean()) * (m2 - m2.mean())).mean().item()) < 2e-3                 # neighbouring seeds independent
    assert ops.dropout_keep_mask(1000, 0.0, 5).all()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layernorm_dropout(dt):
    M, N, p_in, p_out, s_in, s_out = 333, 768, 0.1, 0.25, 11, 99
    g = torch.Generator().manual_seed(0)
    x, res = torch.randn(M, N, generator=g).to(DEV).to(dt), torch.randn(M, N, generator=g).to(DEV).to(dt)
    bias = torch.randn(N, generator=g).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    m_in = ops.dropout_keep_mask(M * N, p_in, s_in).view(M, N).double() / (1 - p_in)
    m_out = ops.dropout_keep_mask(M * N, p_out, s_out).view(M, N).double() / (1 - p_out)
    y, z, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-12, bias=bias, res=res, p_in=p_in, seed_in=s_in, p_out=p_out,
                                         seed_out=s_out)
    zr = (x.double() + bias.double()) * m_in + res.double()
    tol = 2e-5 if dt == torch.float32 else 2.5e-2
    assert rel(z, zr) < tol
    zs = z.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(zs, (N,), gamma.double(), beta.double(), 1e-12) * m_out
    assert rel(y, yr) < tol
    dy = torch.randn(M, N, generator=g).to(DEV).to(dt)
    (yr * dy.double()).sum().backward()
    dgamma, dbeta = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    dbias = torch.zeros(N, device=DEV)
    dz, dzd = ops.layernorm_bwd(dy, None, z, mean, rstd, gamma, dgamma, dbeta, p_in=p_in, seed_in=s_in, p_out=p_out,
                                seed_out=s_out, dbias=dbias)
    assert rel(dbias, dzd.double().sum(0)) < 1e-5
    assert rel(dz, zs.grad) < tol
    assert rel(dzd, zs.grad * m_in) < tol


@pytest.mark.parametrize("T", [30, 50, 100])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_attention_dropout(dt, T):
    """T = 50: the 64 x 64 tile (abstracts / bodies, T/parameters.py:43-44) -- exact VALU kernels in fp32, attention_mfma64.hip in the 16-bit
    modes -- whose mask stream is indexed with a row pitch of 64."""
    n_seq, nh, dh, p, seed = 4, 3, 64, 0.2, 777
    TP = 32 if T <= 32 else (64 if T <= 64 else 32 * ((T + 31) // 32))      # (T > 64: the row-strip kernels, mask pitch 32 * ceil(T / 32))
    H = nh * dh
    g = torch.Generator().manual_seed(1)
    qkv = (0.7 * torch.randn(n_seq * T, 3 * H, generator=g)).to(DEV).to(dt)
    keep = torch.ones(n_seq, T, device=DEV)
    keep[1, 20:] = 0
    scale = 1 / math.sqrt(dh)
    desc = ops.attn_desc(n_seq, T, nh, dh, False, scale, ops.FLT_MIN_MASK, dt, p, seed)
    ctx = ops.attn_fwd(desc, qkv, keep)
    thr = int(p * 65536)
    mask = ops.dropout_keep_mask(n_seq * nh * TP * TP, p, seed).view(n_seq, nh, TP, TP)[:, :, :T, :T].double() / (1 - thr / 65536.0)
    qd = qkv.double().requires_grad_(True)
    q, k, v = (qd[:, i * H:(i + 1) * H].reshape(n_seq, T, nh, dh).transpose(1, 2) for i in range(3))
    att = q @ k.transpose(-1, -2) * scale + torch.where(keep[:, None, None, :] != 0, 0.0, -1e30)
    pr = torch.softmax(att, -1) * mask
    ref = (pr @ v).transpose(1, 2).reshape(n_seq * T, H)
    assert rel(ctx, ref) < (2e-5 if dt == torch.float32 else 2.5e-2)
    dctx = torch.randn(n_seq * T, H, generator=g).to(DEV).to(dt)
    (ref * dctx.double()).sum().backward()
    dqkv = ops.attn_bwd(desc, qkv, keep, dctx)
    assert rel(dqkv, qd.grad) < (1e-4 if dt == torch.float32 else 3e-2)


def test_model_directional_derivative_under_dropout():
    """fp32 path, training mode, frozen dropout stream: (L(w + eps v) - L(w - eps v)) / (2 eps) == <grad, v>."""
    from idvs.morec_amd.model import BertShape, HipBertModel, Model
    from idvs.morec_amd.utils.detgen import det_param
    S, D, T, item_num, B = 6, 64, 30, 40, 6
    shape = BertShape.named("micro")
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                                 num_words_title=T, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                 bert_model_load="bert_micro", word_embedding_dim=64, compute_dtype="fp32")
    rng = np.random.default_rng(3)
    pop = rng.random(item_num + 1) + 0.05
    pop[1:] /= pop[1:].sum()
    pop[0] = 1
    m = Model(args, item_num, True, HipBertModel(shape), pop)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(torch.from_numpy(det_param(k, tuple(v.shape))))
    m = m.to(DEV).train()
    torch.manual_seed(20260927)     # the dropout stream derives from torch.initial_seed(): make the test order-independent
    content = np.zeros((item_num + 1, 2 * T), dtype=np.int64)
    for i in range(1, item_num + 1):
        L = int(rng.integers(3, T + 1))
        content[i, :L] = rng.integers(1, 512, L)
        content[i, T:T + L] = 1
    ids = rng.integers(1, item_num + 1, (B, S + 1))
    ids[0, :3] = 0
    lm = np.ones((B, S), dtype=np.float32)
    lm[0, :3] = 0
    tid, tit, tlm = (torch.from_numpy(ids).to(DEV).view(-1), torch.from_numpy(content[ids.reshape(-1)]).to(DEV),
                     torch.from_numpy(lm).to(DEV))

    def loss_at():
        m._drop_calls = 0    # same dropout stream on every evaluation
        return m(tid, tit, tlm, DEV)

    m.eval()
    l_eval = loss_at().item()
    m.train()
    l0 = loss_at()
    assert abs(l0.item() - l_eval) > 1e-3 and math.isfinite(l0.item())     # dropout really is active
    l0.backward()
    params = [p for n, p in m.named_parameters() if p.grad is not None]
    # probe along the (normalised) gradient itself: the largest first-order signal for the smallest step, which keeps
    # the fp32 finite difference out of both the round-off and the curvature / ReLU-kink regimes
    gnorm = math.sqrt(sum((p.grad.double() ** 2).sum().item() for p in params))
    vs = [(p.grad / gnorm).clone() for p in params]
    dot = gnorm
    eps = 1e-2
    with torch.no_grad():
        for p, v in zip(params, vs):
            p.add_(eps * v)
        lp = loss_at().item()
        for p, v in zip(params, vs):
            p.sub_(2 * eps * v)
        lm_ = loss_at().item()
    fd = (lp - lm_) / (2 * eps)
    print(f"directional derivative: fd {fd:.6f} vs analytic {dot:.6f}; eval loss {l_eval:.4f} train loss {l0.item():.4f}")
    assert abs(fd - dot) < 2e-2 * max(1.0, abs(dot))
