"""Kernel-level parity on a real MI355X: every C-ABI entry point against a plain PyTorch fp64/fp32
reference of the same op (and the CPU oracle for the loss / ranks).  Tolerances: exact-fp32 MFMA path
1e-5-class; bf16 path 2e-2 relative to the tensor's max-abs (stated per test)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from idvs.morec_amd import ops  # noqa: E402
from idvs.morec_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU  # noqa: E402

DEV = "cuda"
DT = [torch.float32, torch.bfloat16, torch.float16]


def tol(dt):
    """fp16 (11-bit significand) is held to an eighth of the bf16 (8-bit) bound: the three extra bits are the point of the mode."""
    return 2e-5 if dt == torch.float32 else (2.5e-2 if dt == torch.bfloat16 else 3.2e-3)


def t16(dt, f32, bf16):
    """Tolerance by storage type: ``f32`` for the exact path, ``bf16`` for bf16, an eighth of that for fp16."""
    return f32 if dt == torch.float32 else (bf16 if dt == torch.bfloat16 else max(bf16 / 8.0, f32))


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def rnd(*shape, dt=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dt)


def test_probe_mfma_layouts():
    out = ops.probe().cpu().numpy()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/probe.txt", "w") as f:
        f.write("bf16 16x16x32:\n%s\nf32 16x16x4:\n%s\ntr_b16 (lane: 4 values):\n%s\n" % (
            out[:256].reshape(64, 4), out[256:512].reshape(64, 4), out[512:768].reshape(64, 4)))
    lanes = np.arange(64)
    exp_bf = np.stack([16 * ((lanes >> 4) * 4 + r) + (lanes & 15) for r in range(4)], 1)
    assert np.array_equal(out[:256].reshape(64, 4), exp_bf)
    exp_f = np.where(((lanes >> 4) * 4 + np.arange(4)[:, None]) < 4,
                     16 * ((lanes >> 4) * 4 + np.arange(4)[:, None]) + (lanes & 15), 0).T
    assert np.array_equal(out[256:512].reshape(64, 4), exp_f)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(128, 128, 64), (300, 260, 136), (77, 512, 768), (2560, 2688, 512), (4096, 768, 3072)])
def test_gemm_plain(dt, shape):
    M, N, K = shape
    a, b = rnd(M, K, dt=dt), rnd(N, K, dt=dt, seed=1)
    c = ops.gemm_nt(a, b)
    ref = a.double() @ b.double().t()
    assert rel(c, ref) < tol(dt)


@pytest.mark.parametrize("dt", DT)
def test_gemm_epilogues(dt):
    M, N, K = 260, 384, 256
    a, b = rnd(M, K, dt=dt, scale=0.2), rnd(N, K, dt=dt, scale=0.2, seed=1)
    bias = rnd(N, seed=2)
    aux = torch.empty(M, N, device=DEV, dtype=dt)
    c = ops.gemm_nt(a, b, bias=bias, act=ACT_GELU, aux_out=aux)
    pre = a.double() @ b.double().t() + bias.double()
    assert rel(aux, pre) < tol(dt)
    assert rel(c, torch.nn.functional.gelu(pre)) < tol(dt)
    c = ops.gemm_nt(a, b, bias=bias, act=ACT_RELU)
    assert rel(c, torch.relu(pre)) < tol(dt)
    # derivative epilogues
    u = rnd(M, N, dt=dt, seed=3)
    c = ops.gemm_nt(a, b, dact=ACT_GELU, dact_in=u)
    ud = u.double().requires_grad_(True)
    torch.nn.functional.gelu(ud).sum().backward()
    assert rel(c, (a.double() @ b.double().t()) * ud.grad) < tol(dt)
    c = ops.gemm_nt(a, b, dact=ACT_RELU, dact_in=u)
    assert rel(c, (a.double() @ b.double().t()) * (u.double() > 0)) < tol(dt)
    # accumulate and alpha
    c0 = rnd(M, N, dt=dt, seed=4)
    c = ops.gemm_nt(a, b, out=c0.clone(), accumulate=1, alpha=0.5)
    assert rel(c, c0.double() + 0.5 * (a.double() @ b.double().t())) < tol(dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(260, 384, 256), (5000, 3072, 768)])
@pytest.mark.parametrize("act", [ACT_GELU, ACT_RELU])
def test_gemm_activation_derivative_pair(dt, shape, act):
    """``aux_deriv`` forward + ``MOREC_DACT_MUL`` backward (how the FFN of every encoder layer runs: T/model/modules.py:14-17, HF
    BertIntermediate): the forward GEMM leaves act'(pre) as its second output, the backward GEMM multiplies by it -- together the
    autograd backward of ``act(x W^T + b)``.  Small shape -> the two-buffer kernel, large -> the eight-phase kernel (bf16)."""
    from idvs.morec_amd._lib import DACT_MUL
    M, N, K = shape
    a, b = rnd(M, K, dt=dt, scale=0.2), rnd(N, K, dt=dt, scale=0.2, seed=1)
    bias = rnd(N, seed=2)
    dprime = torch.empty(M, N, device=DEV, dtype=dt)
    c = ops.gemm_nt(a, b, bias=bias, act=act, aux_out=dprime, aux_deriv=True)
    pre = (a.double() @ b.double().t() + bias.double()).requires_grad_(True)
    y = torch.nn.functional.gelu(pre) if act == ACT_GELU else torch.relu(pre)
    y.sum().backward()
    assert rel(c, y.detach()) < tol(dt)
    if act == ACT_GELU:
        assert float((dprime.double() - pre.grad).abs().max()) < t16(dt, 1e-5, 1e-2)
    else:     # 0 / 1 except where the pre-activation rounds across zero
        assert float(((dprime.double() - pre.grad).abs() > 0).double().mean()) < t16(dt, 1e-6, 5e-3)
    # backward GEMM of the pair: dU = (dZ . W2) * act'  with the column sums (d b1) fused
    dz, w2 = rnd(M, 64, dt=dt, scale=0.2, seed=7), rnd(N, 64, dt=dt, scale=0.2, seed=8)
    cs = torch.zeros(N, device=DEV)
    du = ops.gemm_nt(dz, w2, dact=DACT_MUL, dact_in=dprime, colsum_out=cs)
    ref = (dz.double() @ w2.double().t()) * dprime.double()
    assert rel(du, ref) < tol(dt)
    assert float((cs.double() - du.double().sum(0)).abs().max()) < 1e-5 * float(du.double().sum(0).abs().max()) + 1e-6 * M ** 0.5
    # and the same product through the explicit-derivative epilogue (dact = act, operand = pre-activation) agrees
    pre_t = torch.empty(M, N, device=DEV, dtype=dt)
    ops.gemm_nt(a, b, bias=bias, act=act, aux_out=pre_t)
    du2 = ops.gemm_nt(dz, w2, dact=act, dact_in=pre_t)
    assert rel(du, du2.double()) < t16(dt, 1e-5, 2e-2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(260, 384, 256), (77, 96, 64), (3000, 3072, 768), (5000, 136, 96), (2100, 1536, 384)])
def test_gemm_dact_fused_colsum(dt, shape):
    """``morec_gemm_nt_colsum``: same C as the plain call (bit for bit), and colsum_out += C.sum(0) -- the bias gradient that
    autograd's ``dY.sum(0)`` gives for the layer below -- on both epilogue forms (wave slices N <= 1024, block staging above),
    ragged M / N tiles included.  Accumulates into the caller's buffer."""
    M, N, K = shape
    a, b = rnd(M, K, dt=dt, scale=0.2), rnd(N, K, dt=dt, scale=0.2, seed=1)
    u = rnd(M, N, dt=dt, seed=3)
    for act in (ACT_GELU, ACT_RELU):
        c0 = ops.gemm_nt(a, b, dact=act, dact_in=u)
        init = rnd(N, seed=5)
        cs = init.clone()
        c1 = ops.gemm_nt(a, b, dact=act, dact_in=u, colsum_out=cs)
        assert torch.equal(c0, c1)
        ref = init.double() + c0.double().sum(0)
        assert float((cs.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-6 * M ** 0.5
    with pytest.raises(RuntimeError):       # only behind an activation-derivative epilogue
        ops.gemm_nt(a, b, colsum_out=torch.zeros(N, device=DEV))


@pytest.mark.parametrize("dt", DT)
def test_gemm_splitk_atomic(dt):
    M, N, K = 768, 768, 20160
    a, b = rnd(M, K, dt=dt, scale=0.1), rnd(N, K, dt=dt, scale=0.1, seed=1)
    out = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    ops.gemm_nt(a, b, out=out, accumulate=2, split_k=8)
    assert rel(out, a.double() @ b.double().t()) < tol(dt)
    # odd K tail + pitch larger than K
    K2 = 1000
    out.zero_()
    ops.gemm_nt(a, b, out=out, accumulate=2, split_k=3, K=K2)
    assert rel(out, a[:, :K2].double() @ b[:, :K2].double().t()) < tol(dt)


@pytest.mark.parametrize("shape", [(64, 256, 256, 1), (200, 96, 264, 1), (4096, 768, 2304, 4), (20160, 768, 768, 9),
                                   (2590, 1536, 512, 5), (77, 8, 16, 1)])
@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16])
def test_gemm_tn(shape, dt16):
    """dW = dY^T X straight from the row-major operands (transpose reads), incl. token / column tails and split-m atomics."""
    M, N, K, split = shape
    dy, x = rnd(M, N, dt=dt16, scale=0.2), rnd(M, K, dt=dt16, scale=0.2, seed=1)
    out = torch.zeros(N, K, device=DEV)
    ops.gemm_tn_(dy, x, out, split_m=split)
    ref = dy.double().t() @ x.double()
    assert rel(out, ref) < 2e-5 * math.sqrt(M) + 1e-6
    out3 = torch.ones(N, K, device=DEV)                      # atomics path accumulates into what is there
    ops.gemm_tn_(dy, x, out3, split_m=split, slabs=False)
    assert rel(out3 - 1, ref) < 2e-5 * math.sqrt(M) + 1e-5
    out4 = torch.ones(N, K, device=DEV)                      # slab path with accumulate: C += sum of slabs
    ops.gemm_tn_(dy, x, out4, split_m=split)
    assert rel(out4 - 1, ref) < 2e-5 * math.sqrt(M) + 1e-5
    if split == 1:
        out2 = torch.full((N, K), 7.0, device=DEV)
        ops.gemm_tn_(dy, x, out2, split_m=1, accumulate=False)
        assert torch.equal(out2, out)


@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,bias", [(2560, 512, 512, False), (2560, 512, 2048, True), (640, 2048, 2048, True), (2688, 512, 512, False), (77, 72, 64, True),
                                        (1000, 260, 520, False), (64, 64, 4096, True), (3000, 1024, 136, True), (130, 4, 128, False)])
def test_gemm_nt_small_problems(M, N, K, bias, dt16):
    """The SASRec-sized products (2 560 ... 640 rows, D = 512 ... 2048: the two-buffer tile kernel): ragged M / N / K, a strided A and C, bias,
    against the exact product of the same 16-bit operands; nothing written outside the M x N block."""
    a_full = rnd(M, K + 24, dt=dt16, scale=0.3)
    a = a_full[:, :K]
    b = rnd(N, K, dt=dt16, scale=0.3, seed=1)
    bv = rnd(N, seed=2) if bias else None
    guard = torch.full((M + 3, N + 8), 5.0, device=DEV, dtype=dt16)
    out = guard[:M, :N]
    ops.gemm_nt(a_full, b, out=guard, bias=bv, M=M, N=N, K=K, lda=a_full.stride(0), ldc=guard.stride(0))      # (whole tensors + explicit extents)
    ref = a.double() @ b.double().t() + (bv.double() if bias else 0.0)
    ulp = 2.0 ** (-8 if dt16 == torch.bfloat16 else -11)
    bound = ulp * ref.abs() + 3e-6 * math.sqrt(K) + 1e-30
    assert bool(((out.double() - ref).abs() <= bound).all()), float(((out.double() - ref).abs() / bound).max())
    assert bool((guard[M:] == 5.0).all()) and bool((guard[:, N:] == 5.0).all())


@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(120000, 96, 384), (120000, 384, 96), (90001, 288, 96), (90000, 96, 96), (70000, 192, 768), (70000, 576, 192),
                                   (60000, 96, 48), (50000, 128, 512), (50000, 1152, 384)])
def test_gemm_tn_outputs_narrower_than_the_tile(M, N, K, dt16):
    """Swin's weight-gradient shapes (C = 96 ... 384): 256 x 256 tiles of which whole 32 x 32 blocks lie outside the matrix -- the
    eight-phase kernel's block-skipping variant (gemm_tn8p.hip, SKIP) must leave the blocks inside untouched and write nothing outside."""
    from idvs.morec_amd import engine
    dy, x = rnd(M, N, dt=dt16, scale=0.2), rnd(M, K, dt=dt16, scale=0.2, seed=1)
    split = engine._splitk(N, K, M)
    guard = torch.full((N + 8, K + 8), 3.0, device=DEV)
    out = guard[:N, :K]
    ops.gemm_tn_(dy, x, out, split_m=split, accumulate=False)
    ref = dy.double().t() @ x.double()
    assert rel(out, ref) < 2e-5 * math.sqrt(M) + 1e-6
    assert bool((guard[N:] == 3.0).all()) and bool((guard[:, K:] == 3.0).all())
    out2 = torch.ones(N, K, device=DEV)
    ops.gemm_tn_(dy, x, out2, split_m=split)                 # accumulate into what is there
    assert rel(out2 - 1, ref) < 2e-5 * math.sqrt(M) + 1e-5


@pytest.mark.parametrize("dt", DT)
def test_transpose_cast_colsum(dt):
    x = rnd(300, 170, dt=dt)
    xt = ops.transpose(x)
    assert xt.shape == (170, 304) and torch.equal(xt[:, :300], x.t()) and (xt[:, 300:] == 0).all()
    w = rnd(96, 200)
    for d16 in (torch.bfloat16, torch.float16):
        wt = ops.transpose(w, out_dtype=d16)
        assert torch.equal(wt[:, :96], w.t().to(d16))
        assert torch.equal(ops.cast(w, d16), w.to(d16))
    x4 = rnd(1300, 172, dt=dt, seed=3)
    out = torch.zeros(172, device=DEV)
    ops.colsum_(x4, out)
    assert rel(out, x4.double().sum(0)) < 1e-5
    xt4 = ops.transpose(x4)          # vectorised path (C % 4 == 0), R % 8 != 0 -> zero pad
    assert xt4.shape == (172, 1304) and torch.equal(xt4[:, :1300], x4.t()) and (xt4[:, 1300:] == 0).all()
    # many matrices, one launch (the per-step W^T refresh): same results as the single-matrix entry point, pads untouched
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (100, 64), (1300, 172), (8, 4)]
    srcs = [rnd(r, c, dt=dt, seed=20 + i) for i, (r, c) in enumerate(shapes)]
    dsts = [torch.zeros((c, (r + 7) // 8 * 8), device=DEV, dtype=dt) for r, c in shapes]
    tb = ops.TransposeBatch(list(zip(srcs, dsts)))
    tb.run()
    for (r, c), x_, y_ in zip(shapes, srcs, dsts):
        assert torch.equal(y_[:, :r], x_.t()) and (y_[:, r:] == 0).all()
        assert torch.equal(y_, ops.transpose(x_))
    assert not ops.TransposeBatch.eligible(rnd(10, 6, dt=dt), torch.zeros(6, 16, device=DEV, dtype=dt))     # C % 4 != 0: single-matrix path


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("N", [64, 512, 768, 2048])
def test_layernorm(dt, N):
    M, S = 203 * 4 + 1, 7
    M = (M // S) * S
    x, res = rnd(M, N, dt=dt), rnd(M, N, dt=dt, seed=1)
    bias, pos = rnd(N, seed=2), rnd(S, N, seed=3)
    gamma, beta = 1 + 0.1 * rnd(N, seed=4), 0.1 * rnd(N, seed=5)
    y, z, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6, bias=bias, res=res, pos=pos, pos_period=S)
    zr = x.double() + bias.double() + res.double() + pos.double().repeat(M // S, 1)
    assert rel(z, zr) < tol(dt)
    zs = z.double().requires_grad_(True)  # the kernel normalises the stored (rounded) z
    yr = torch.nn.functional.layer_norm(zs, (N,), gamma.double(), beta.double(), 1e-6)
    assert rel(y, yr) < tol(dt)
    dy_a, dy_b = rnd(M, N, dt=dt, seed=6), rnd(M, N, dt=dt, seed=7)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(zs, (N,), gd, bd, 1e-6)
    (yr * (dy_a.double() + dy_b.double())).sum().backward()
    dgamma, dbeta = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    dbias = torch.zeros(N, device=DEV)
    dz, dzd = ops.layernorm_bwd(dy_a, dy_b, z, mean, rstd, gamma, dgamma, dbeta, dbias=dbias)
    assert dzd is dz
    assert rel(dbias, dz.double().sum(0)) < 1e-5     # fused gradient of the pre-add bias
    assert rel(dz, zs.grad) < tol(dt)
    assert rel(dgamma, gd.grad) < 1e-4 and rel(dbeta, bd.grad) < 1e-4
    dpos = torch.zeros(S, N, device=DEV)
    ops.pos_grad_(dz, dpos, S)
    assert rel(dpos, dz.double().view(M // S, S, N).sum(0)) < 1e-5
    # no pre-add: z aliases x
    y2, z2, _, _ = ops.layernorm_fwd(x, gamma, beta, 1e-12)
    assert z2 is x
    assert rel(y2, torch.nn.functional.layer_norm(x.double(), (N,), gamma.double(), beta.double(), 1e-12)) < tol(dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("nseq,S,N", [(128, 20, 512), (64, 10, 2048), (3, 7, 64), (1, 5, 128), (700, 3, 768), (9, 4, 4104), (33, 6, 516)])
def test_pos_grad_shapes(dt, nseq, S, N):
    """dpos[m % S] += dz[m] at the recommender's sizes (text D = 512, vision D = 2048), a single sequence, many sequences with few
    positions, and row widths outside the one-vector-per-thread layout (4104 > 256 vectors; 516 is not a multiple of 8: scalar kernel);
    accumulates into what is there."""
    dz = rnd(nseq * S, N, dt=dt, seed=N + S)
    dpos = torch.full((S + 2, N), 0.5, device=DEV)
    ops.pos_grad_(dz, dpos, S)
    assert rel(dpos[:S] - 0.5, dz.double().view(nseq, S, N).sum(0)) < 1e-5 and (dpos[S:] == 0.5).all()


def attn_ref(qkv, keep, n_heads, causal, scale, mask_value):
    M, H3 = qkv.shape
    n_seq, T = keep.shape
    H = H3 // 3
    dh = H // n_heads
    q, k, v = (qkv[:, i * H:(i + 1) * H].reshape(n_seq, T, n_heads, dh).transpose(1, 2) for i in range(3))
    kept = (keep != 0)[:, None, None, :].expand(n_seq, 1, T, T)
    if causal:
        kept = torch.tril(kept)
    # same fp32 absorb semantics as the reference, evaluated in float32 on purpose
    att = (q.float() @ k.float().transpose(-1, -2)) * scale + torch.where(kept, 0.0, mask_value).float()
    p = torch.softmax(att.double(), -1)
    return (p @ v.double()).transpose(1, 2).reshape(M, H)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", [(5, 30, 12, 64, False, ops.FLT_MIN_MASK), (4, 20, 2, 256, True, -1e9),
                                 (3, 7, 2, 32, True, -1e9), (6, 32, 2, 64, False, -10000.0), (2, 10, 2, 1024, True, -1e9),
                                 # 32 < T <= 64 (abstracts / bodies of 50 tokens, T/parameters.py:43-44; longer behaviour sequences): the 64 x 64
                                 # tile -- exact VALU kernels in fp32, attention_mfma64.hip (two k-steps over keys / queries) in the 16-bit modes
                                 (3, 50, 12, 64, False, ops.FLT_MIN_MASK), (2, 48, 2, 256, True, -1e9), (4, 64, 4, 64, True, -1e9),
                                 (5, 33, 2, 32, False, ops.FLT_MIN_MASK), (3, 40, 2, 128, True, -1e9), (130, 50, 12, 64, False, ops.FLT_MIN_MASK),
                                 # 64 < T <= 256 (longer than any launcher of the reference sets, accepted by its command line): the row-strip
                                 # kernels of attention.hip in every dtype
                                 (3, 100, 2, 64, True, -1e9), (2, 128, 12, 64, False, ops.FLT_MIN_MASK), (2, 200, 2, 256, True, -1e9),
                                 (1, 256, 2, 32, False, ops.FLT_MIN_MASK), (4, 65, 2, 128, True, -1e9)])
def test_attention(dt, cfg):
    n_seq, T, nh, dh, causal, mv = cfg
    H = nh * dh
    qkv = rnd(n_seq * T, 3 * H, dt=dt, scale=0.7)
    keep = torch.ones(n_seq, T, device=DEV)
    for s in range(n_seq):
        n_pad = (s * 5) % T
        if causal:
            keep[s, :n_pad] = 0        # left padding (log_mask)
        else:
            keep[s, T - n_pad:] = 0    # right padding (attention_mask)
    if not causal:
        keep[n_seq - 1, :] = 0         # all-PAD title: fully masked rows
    scale = 1.0 / math.sqrt(dh)
    desc = ops.attn_desc(n_seq, T, nh, dh, causal, scale, mv, dt)
    ctx = ops.attn_fwd(desc, qkv, keep)
    qd = qkv.double().requires_grad_(True)
    ref = attn_ref(qd, keep, nh, causal, scale, mv)
    assert rel(ctx, ref) < tol(dt)
    dctx = rnd(n_seq * T, H, dt=dt, seed=9)
    (ref * dctx.double()).sum().backward()
    dqkv = ops.attn_bwd(desc, qkv, keep, dctx)
    assert rel(dqkv, qd.grad) < t16(dt, 1e-4, 3e-2)
    # same backward with the fused q|k|v bias gradient: dqkv bit-identical, dbias += column sums of dqkv as stored
    dbias = torch.full((3 * H,), 0.25, device=DEV)
    dqkv2 = ops.attn_bwd(desc, qkv, keep, dctx, dbias=dbias)
    assert torch.equal(dqkv2, dqkv)
    want = 0.25 + dqkv.double().sum(0)
    # fp32: sums of the stored rows; bf16 MFMA path: fp32 sums of the rows before their bf16 rounding (rounding noise apart)
    assert (dbias.double() - want).abs().max().item() <= t16(dt, 1e-5, 3e-3) * max(1.0, want.abs().max().item())
    assert rel(dbias - 0.25, qd.grad.sum(0)) < t16(dt, 1e-4, 3e-2)


@pytest.mark.parametrize("dt", DT)
def test_bert_embed(dt):
    n_seq, T, H, V = 37, 30, 768, 1000
    ids = torch.randint(0, V, (n_seq * T,), device=DEV, dtype=torch.int32)
    ids[::7] = 0
    word, pos, typ = rnd(V, H, scale=0.05), rnd(64, H, scale=0.05, seed=1), rnd(2, H, scale=0.05, seed=2)
    gamma, beta = 1 + 0.1 * rnd(H, seed=3), 0.1 * rnd(H, seed=4)
    y, z, mean, rstd = ops.bert_embed_fwd(ids, word, pos, typ[0].contiguous(), gamma, beta, 1e-12, T, dt)
    zr = word.double()[ids.long()] + pos.double()[:T].repeat(n_seq, 1) + typ.double()[0]
    assert rel(z, zr) < tol(dt)
    assert rel(y, torch.nn.functional.layer_norm(z.double(), (H,), gamma.double(), beta.double(), 1e-12)) < tol(dt)
    dz = rnd(n_seq * T, H, dt=dt, seed=5)
    dword, dpos, dtyp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros(H, device=DEV)
    ops.bert_embed_bwd_(ids, dz, dword, dpos, dtyp, 0, T)
    ref = torch.zeros_like(word, dtype=torch.float64)
    ref.index_add_(0, ids.long(), dz.double())
    ref[0] = 0
    assert rel(dword, ref) < 1e-5
    assert rel(dpos[:T], dz.double().view(n_seq, T, H).sum(0)) < 1e-5 and (dpos[T:] == 0).all()
    assert rel(dtyp, dz.double().sum(0)) < 1e-5
    # run-length scatter over token-id order (hot tokens: every 5th row is the same id, like [CLS] / [SEP])
    ids2 = ids.clone()
    ids2[::5] = 101
    dword2, dpos2, dtyp2 = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros(H, device=DEV)
    ops.bert_embed_bwd_(ids2, dz, dword2, dpos2, dtyp2, 0, T, torch.argsort(ids2).to(torch.int32))
    ref2 = torch.zeros_like(word, dtype=torch.float64)
    ref2.index_add_(0, ids2.long(), dz.double())
    ref2[0] = 0
    assert rel(dword2, ref2) < 1e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n_seq,T,H", [(150, 30, 768), (131, 7, 128), (70, 5, 516), (3, 30, 1024)])
def test_bert_embed_bwd_shapes(dt, n_seq, T, H):
    """Word / position / type gradients at more shapes: several sequence chunks per position (n_seq > 64), a last wave with fewer
    than 32 sorted rows, rows wider / narrower than one 256-lane pass, and a bf16 row width that is not a multiple of the 16-byte
    vector (516: the scalar position kernel); sorted (run-length) and plain scatter against an fp64 index_add."""
    V = 300
    g = torch.Generator(device=DEV).manual_seed(n_seq * 1000 + H)
    ids = torch.randint(0, V, (n_seq * T,), device=DEV, dtype=torch.int32, generator=g)
    ids[::3] = 0
    ids[1::4] = 17
    dz = rnd(n_seq * T, H, dt=dt, seed=H)
    ref = torch.zeros((V, H), device=DEV, dtype=torch.float64)
    ref.index_add_(0, ids.long(), dz.double())
    ref[0] = 0
    for order in (None, torch.argsort(ids, stable=True).to(torch.int32)):
        dword, dpos, dtyp = torch.zeros((V, H), device=DEV), torch.zeros((T + 3, H), device=DEV), torch.zeros(H, device=DEV)
        ops.bert_embed_bwd_(ids, dz, dword, dpos, dtyp, 0, T, order)
        assert rel(dword, ref) < 1e-5
        assert rel(dpos[:T], dz.double().view(n_seq, T, H).sum(0)) < 1e-5 and (dpos[T:] == 0).all()
        assert rel(dtyp, dz.double().sum(0)) < 1e-5


@pytest.mark.parametrize("dt", DT)
def test_indexed_rows_negative_gather_index_is_a_zero_row(dt):
    src = rnd(50, 136, dt=dt)
    idx = torch.randint(0, 50, (333,), device=DEV, dtype=torch.int32)
    idx[::3] = -1
    out = ops.indexed_rows_copy(src, torch.full((333, 136), 7.0, device=DEV, dtype=dt), in_idx=idx)
    want = src[idx.clamp(min=0).long()]
    want[idx < 0] = 0
    assert torch.equal(out, want)


@pytest.mark.parametrize("dt", DT)
def test_gather_scatter(dt):
    V, D, R = 500, 512, 333
    table = rnd(V, D)
    idx = torch.randint(0, V, (R,), device=DEV, dtype=torch.int32)
    idx[::5] = 0
    out = ops.gather_rows(table, idx, dt)
    assert torch.equal(out, table[idx.long()].to(dt))
    d = rnd(R, D, dt=dt, seed=1)
    dt_ = torch.zeros_like(table)
    ops.scatter_add_rows_(d, idx, dt_, 0)
    ref = torch.zeros_like(table, dtype=torch.float64)
    ref.index_add_(0, idx.long(), d.double())
    ref[0] = 0
    assert rel(dt_, ref) < 1e-5
    hid = rnd(40 * 30, 768, dt=dt, seed=2)
    cls = torch.empty(40, 768, device=DEV, dtype=dt)
    ops.strided_rows_copy(hid, cls, 40, 768, 30, 1)
    assert torch.equal(cls, hid.view(40, 30, 768)[:, 0])
    back = torch.zeros_like(hid)
    ops.strided_rows_copy(cls, back, 40, 768, 1, 30)
    assert torch.equal(back.view(40, 30, 768)[:, 0], cls) and (back.view(40, 30, 768)[:, 1:] == 0).all()


def _ce_case(B, S, D, item_num, seed, n_ranks=1, rank=0):
    rng = np.random.default_rng(seed)
    Bt = B * n_ranks
    ids = np.zeros((Bt, S + 1), dtype=np.int64)
    lm = np.zeros((Bt, S), dtype=np.float32)
    for b in range(Bt):
        L = int(rng.integers(2, S + 2))
        seq = rng.integers(1, item_num + 1, L)
        if L > 3:
            seq[-2] = seq[0]
        ids[b, S + 1 - L:] = seq
        lm[b, S + 1 - L:] = 1
    pop = rng.random(item_num + 1) + 0.01
    pop[1:] /= pop[1:].sum()
    pop[0] = 1.0
    return ids, lm, pop


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", [(6, 5, 64, 30, 1, 0), (16, 20, 512, 200, 1, 0), (128, 20, 512, 5000, 1, 0),
                                 (8, 20, 128, 100, 4, 2), (5, 10, 2048, 60, 2, 1)])
def test_inbatch_ce(dt, cfg):
    import morec_oracle as orc
    from morec_oracle import bookkeeping as bk
    B, S, D, item_num, n_ranks, rank = cfg
    ids_all, lm_all, pop = _ce_case(B, S, D, item_num, seed=B + S, n_ranks=n_ranks)
    Nc = ids_all.size
    ids, lm = ids_all[rank * B:(rank + 1) * B], lm_all[rank * B:(rank + 1) * B]
    E = rnd(Nc, D, dt=dt, scale=0.5)
    P = rnd(B * S, D, dt=dt, scale=0.5, seed=1)
    off = rank * B * (S + 1)
    n_valid = int((lm_all != 0).sum())
    # oracle (CPU, fp64 on the same rounded inputs)
    Pc, Ec = P.double().cpu().requires_grad_(True), E.double().cpu().requires_grad_(True)
    loss_ref = orc.inbatch_ce_loss(Pc, Ec, ids, lm, pop, S, pool_ids=ids_all, pool_log_mask=lm_all, col_offset=off,
                                   n_valid_total=n_valid)
    loss_ref.backward()
    desc = ops.ce_desc(B, S, D, Nc, off, dt)
    ws = ops.ce_workspace(desc, DEV)
    t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(d)
    row_ids, col_ids = t(ids.reshape(-1), torch.int32), t(ids_all.reshape(-1), torch.int32)
    logpop = t(bk.log_pop(pop, ids_all), torch.float32)
    col_valid = t(bk.column_valid(lm_all), torch.uint8)
    row_valid = t(lm.reshape(-1) != 0, torch.uint8)
    loss_sum, lse, row_loss = ops.inbatch_ce_fwd(desc, P, E, row_ids, col_ids, logpop, col_valid, row_valid, ws)
    loss = loss_sum.item() / n_valid
    assert abs(loss - loss_ref.item()) < t16(dt, 2e-5, 2e-2) * max(1.0, abs(loss_ref.item()))
    dP, dE = ops.inbatch_ce_bwd(desc, P, E, row_ids, col_ids, logpop, col_valid, row_valid, lse, None, 1.0 / n_valid, ws)
    assert rel(dP.cpu(), Pc.grad) < t16(dt, 1e-4, 3e-2)
    assert rel(dE.cpu(), Ec.grad) < t16(dt, 1e-4, 3e-2)
    if dt != torch.float32 and Nc % 8 == 0:
        # pooled-negative form: dE handed out in fp32 (it is reduce-scattered over ranks before any rounding)
        desc32 = ops.ce_desc(B, S, D, Nc, off, dt, dE_fp32=True)
        dP2, dE32 = ops.inbatch_ce_bwd(desc32, P, E, row_ids, col_ids, logpop, col_valid, row_valid, lse, None, 1.0 / n_valid, ws)
        assert dE32.dtype == torch.float32 and torch.equal(dP2, dP)
        assert rel(dE32.cpu(), Ec.grad) < 3e-2
        assert float((dE32.to(dt).float() - dE.float()).abs().max()) <= t16(dt, 0, 1e-2) * float(dE.float().abs().max()) + 1e-9


def test_inbatch_ce_golden(golden_dir):
    """ID-tower goldens captured from the reference: bookkeeping through the HIP kernel (exact-fp32 path)."""
    from morec_oracle import bookkeeping as bk
    g = np.load(os.path.join(golden_dir, "g1_g4_id_tower.npz"))
    for case in ["a", "b", "c", "d", "e"]:
        B, S = int(g[f"{case}.B"]), int(g[f"{case}.S"])
        ids, lm, pop = g[f"{case}.ids"], g[f"{case}.log_mask"], g[f"{case}.pop"]
        rows = bk.valid_rows(lm)
        # feed P = one-hot-ish so logits are recoverable?  Instead check the loss with logits reproduced from the
        # golden's masked logits: use E = I-like embedding is not available; the full-model golden test covers values.
        # Here: masked-cell pattern via gradient support -- dlogit == 0 exactly on masked cells.
        D = 128
        Nc = ids.size
        E = rnd(Nc, D, scale=0.3)
        P = rnd(B * S, D, scale=0.3, seed=1)
        desc = ops.ce_desc(B, S, D, Nc, 0, torch.float32)
        ws = ops.ce_workspace(desc, DEV)
        t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(d)
        args = (t(ids.reshape(-1), torch.int32), t(ids.reshape(-1), torch.int32), t(bk.log_pop(pop, ids), torch.float32),
                t(bk.column_valid(lm), torch.uint8), t(lm.reshape(-1) != 0, torch.uint8))
        loss_sum, lse, row_loss = ops.inbatch_ce_fwd(desc, P, E, *args, ws)
        ops.inbatch_ce_bwd(desc, P, E, *args, lse, None, 1.0, ws)
        torch.cuda.synchronize()
        ldc = ops.pad8(Nc)
        dl = ws[: B * S * ldc * 4].view(torch.float32).view(B * S, ldc)[:, :Nc].cpu().numpy()
        masked = g[f"{case}.masked_valid"]
        lab = g[f"{case}.labels_valid"]
        zero = dl[rows] == 0.0
        # the positive's own gradient softmax-1 may round to 0 when every other cell is masked (case c: B = 1)
        zero[np.arange(rows.size), lab] = False
        assert np.array_equal(zero, masked), case   # zero gradient exactly on the reference's -1e4 cells
        assert (dl[rows, lab] <= 0).all()
        inv = np.setdiff1d(np.arange(B * S), rows)
        assert (dl[inv] == 0).all()


def test_adamw():
    import morec_oracle as orc
    n = 4096 * 3 + 8
    p, g = rnd(n), rnd(n, seed=1, scale=1e-2)
    m, v = rnd(n, seed=2, scale=1e-3), rnd(n, seed=3, scale=1e-3).abs()
    pc, gc, mc, vc = (x.cpu().clone() for x in (p, g, m, v))
    shadow = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    ops.adamw_(p, g, m, v, shadow, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3)
    orc.adamw_step(pc, gc, mc, vc, 3, 1e-4, 0.01)
    assert (p.cpu() - pc).abs().max() < 5e-7 and rel(m.cpu(), mc) < 1e-6 and rel(v.cpu(), vc) < 1e-6
    assert torch.equal(shadow, p.to(torch.bfloat16))


@pytest.mark.parametrize("sdt", [torch.bfloat16, torch.float16])
def test_adamw_step_block_is_the_grad_scaler_protocol(sdt):
    """``morec_step_params`` + ``morec_adamw_sp`` (T/run.py:210,243-247: GradScaler + AdamW): scaled gradients are unscaled by the kernel,
    bias corrections follow the count of APPLIED steps, a non-finite gradient skips the whole update and halves the scale, the scale
    doubles after ``growth_interval`` clean steps -- checked against the CPU oracle's AdamW and torch's GradScaler arithmetic."""
    import morec_oracle as orc
    n = 4096 * 2 + 8
    p, g = rnd(n), rnd(n, seed=1, scale=1e-2)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pc, mc, vc = p.cpu().clone(), m.cpu().clone(), v.cpu().clone()
    shadow = torch.empty(n, device=DEV, dtype=sdt)
    sp = ops.StepParams(DEV, init_scale=1024.0, growth_interval=2)
    scale, applied = 1024.0, 0
    for it in range(5):
        gs = g * scale                                   # what a backward pass that started from loss_scale / n_valid leaves
        if it == 1:
            gs = gs.clone()
            gs[77] = float("inf")
        assert float(sp.loss_scale_dev.item()) == scale
        sp.check_finite_(gs)
        sp.decide_(0.9, 0.999)
        before = p.clone()
        ops.adamw_sp_(p, gs, m, v, shadow, 1e-3, 0.9, 0.999, 1e-8, 0.01, sp)
        h = sp.host()
        if it == 1:                                      # skipped: nothing moves, the scale halves, the step count stays
            assert torch.equal(p, before) and h.apply == 0 and h.skipped == 1 and h.step == applied
            scale *= 0.5
        else:
            applied += 1
            orc.adamw_step(pc, g.cpu(), mc, vc, applied, 1e-3, 0.01)
            assert h.apply == 1 and h.step == applied
            assert (p.cpu() - pc).abs().max() < 1e-6 and rel(m.cpu(), mc) < 1e-5 and rel(v.cpu(), vc) < 5e-5      # (g * S) / S rounds twice
            assert torch.equal(shadow, p.to(sdt))
        assert h.found_inf == 0
        # growth: two clean steps in a row double the scale (the skip reset the tracker)
        if it in (3,):
            scale *= 2.0
    assert float(sp.loss_scale_dev.item()) == scale


def test_eval_rank():
    import morec_oracle as orc
    U, I, D, Hmax = 300, 1000, 64, 12
    prec, emb = rnd(U, D), rnd(I + 1, D, seed=1)
    rng = np.random.default_rng(0)
    hist = np.full((U, Hmax), -1, dtype=np.int32)
    histories, targets = [], np.zeros(U, dtype=np.int64)
    for u in range(U):
        L = int(rng.integers(1, Hmax + 1))
        h = rng.integers(1, I + 1, L)
        if L > 2:
            h[1] = h[0]
        hist[u, :L] = h
        histories.append(h)
        targets[u] = int(rng.integers(1, I + 1)) if u % 17 else int(h[0])  # some targets sit in the history
    rank = ops.eval_rank(prec, emb, torch.from_numpy(hist).to(DEV), torch.from_numpy(targets.astype(np.int32)).to(DEV))
    scores = (prec.double() @ emb.double().t()).float().cpu().numpy()
    ref = orc.eval_ranks(scores, histories, targets)
    ok = targets != np.array([h[0] for h in histories])
    assert np.array_equal(rank.cpu().numpy()[ok], ref[ok])
    assert (rank.cpu().numpy()[~ok] > 10).all()


def test_gemm8p_tail_split_opt_in():
    """The K split of the eight-phase GEMM's tail round (opt-in tuning key): as close to an fp32 product as the unsplit launch, and
    bit-identical from run to run (600 tiles on 256 CUs: 88 tail tiles, each summed by two workgroups)."""
    from idvs.morec_amd import _lib
    L = _lib.lib()
    M, N, K = 51200, 768, 3072
    a = rnd(M, K, dt=torch.bfloat16, scale=0.5); b = rnd(N, K, dt=torch.bfloat16, scale=0.5, seed=3)
    try:
        assert L.morec_tuning_set(b"gemm8p_tail_split", 0) == 0
        base = ops.gemm_nt(a, b)
        assert L.morec_tuning_set(b"gemm8p_tail_split", 1) == 0
        o1, o2 = ops.gemm_nt(a, b), ops.gemm_nt(a, b)
    finally:
        L.morec_tuning_set(b"gemm8p_tail_split", 0)
    assert torch.equal(o1, o2)
    exact = a.float() @ b.float().t()
    e_split, e_base = (o1.float() - exact).abs(), (base.float() - exact).abs()
    assert float(e_split.max()) <= float(e_base.max()) * 1.001 + 1e-6
    assert abs(float(e_split.mean()) / float(e_base.mean()) - 1.0) < 1e-3
    assert int((o1 != base).sum()) < 1e-3 * o1.numel()          # the same numbers up to the summation order
