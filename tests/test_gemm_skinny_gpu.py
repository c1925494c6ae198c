"""-m gpu: the streaming NT GEMM for narrow outputs (``csrc/gemm_skinny.hip``: N <= 128, K <= 384, M >= 8192 -- the Swin stage-1 / stage-2
products, HF modeling_swin.py SwinSelfOutput / SwinOutput / SwinPatchEmbeddings and their autograd backward) against the exact fp64
product of the same 16-bit operands: elementwise within half an output ulp + fp32 accumulation noise; ragged M, K that is not a multiple
of 32 (zero-filled k-steps), strided operands / output, and bit-equality with the tile kernels' result is NOT required (different
summation order), so the bound is the absolute one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(M, N, K, dt, lda=None, ldc=None, seed=0, bias=False):
    from idvs.morec_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed + M + N + K)
    lda, ldc = lda or K, ldc or N
    a_full = (torch.randn(M, lda, generator=g) * 0.5).to(DEV).to(dt)
    b = (torch.randn(N, K, generator=g) * 0.5).to(DEV).to(dt)
    out = torch.full((M, ldc), 7.0, device=DEV, dtype=dt)
    bv = torch.randn(N, generator=g).to(DEV) if bias else None
    ops.gemm_nt(a_full, b, out=out, M=M, N=N, K=K, lda=lda, ldc=ldc, bias=bv)
    ref = a_full[:, :K].double() @ b.double().t() + (bv.double() if bias else 0.0)
    got = out[:, :N].double()
    ulp = 2.0 ** (-8 if dt == torch.bfloat16 else -11)
    bound = ulp * ref.abs() + 3e-6 * np.sqrt(K) + 1e-30
    assert bool(((got - ref).abs() <= bound).all()), float(((got - ref).abs() / bound).max())
    if ldc > N:
        assert bool((out[:, N:] == 7.0).all())          # nothing written past the N columns
    return out


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(300001, 96, 96), (100000, 96, 384), (50017, 96, 288), (70001, 96, 48), (40000, 128, 128), (40003, 128, 384),
                                   (9000, 96, 192), (8192, 64, 64), (33333, 104, 200)])
def test_skinny_products(dt, M, N, K):
    _run(M, N, K, dt)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(300001, 288, 96), (70001, 96, 48), (20000, 192, 64), (50017, 128, 384), (9000, 256, 96)])
def test_skinny_products_with_bias(dt, M, N, K):
    """nn.Linear with its bias: the stage-1 q|k|v projection of Swin-T (N = 288, K = 96) and the patch embedding (K = 48)."""
    _run(M, N, K, dt, bias=True)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_skinny_strided_views_and_switch(dt):
    """A with a row pitch wider than K (a column block of a wider tensor), C with a pitch wider than N; the tuning key routes the same call
    through the tile kernels and both agree to the output rounding."""
    from idvs.morec_amd import _lib, ops
    o1 = _run(20000, 96, 96, dt, lda=288, ldc=160)
    L = _lib.lib()
    assert L.morec_tuning_set(b"gemm_skinny", 1) == 0
    try:
        o2 = _run(20000, 96, 96, dt, lda=288, ldc=160)
    finally:
        L.morec_tuning_set(b"gemm_skinny", 0)
    d = (o1[:, :96].float() - o2[:, :96].float()).abs()
    assert float(d.max()) <= 2.0 ** (-7 if dt == torch.bfloat16 else -10) * float(o2[:, :96].float().abs().max())


# ---- csrc/gemm_skinny_wide.hip: 288 < N <= 512, K <= 128 (Swin stage-1 MLP, HF modeling_swin.py SwinIntermediate / SwinOutput)
def _gelu64(u):
    return 0.5 * u * (1.0 + torch.erf(u / 2.0 ** 0.5))


def _dgelu64(u):
    return 0.5 * (1.0 + torch.erf(u / 2.0 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2.0 * np.pi) ** 0.5


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,gelu", [(300001, 384, 96, True), (100003, 512, 128, True), (8192, 384, 96, True), (20001, 320, 64, True),
                                        (138001, 768, 192, True), (50000, 640, 160, True),
                                        (50000, 448, 128, False), (70001, 384, 96, False)])
def test_wide_bias_gelu(dt, M, N, K, gelu):
    """fc1 + bias + GELU without a second output (streaming kernel; the bias-only cases take the tile kernels); ragged M, N below the padded
    tile width, a pitch wider than N."""
    from idvs.morec_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(DEV).to(dt)
    b = (torch.randn(N, K, generator=g) * 0.3).to(DEV).to(dt)
    bv = torch.randn(N, generator=g).to(DEV)
    ldc = N + 64
    out = torch.full((M, ldc), 7.0, device=DEV, dtype=dt)
    ops.gemm_nt(a, b, out=out, M=M, N=N, K=K, ldc=ldc, bias=bv, act=ops.ACT_GELU if gelu else ops.ACT_NONE)
    u = a.double() @ b.double().t() + bv.double()
    ref = _gelu64(u) if gelu else u
    ulp = 2.0 ** (-8 if dt == torch.bfloat16 else -11)
    bound = ulp * ref.abs() + 3e-6 * np.sqrt(K) + 2e-6          # + the erf polynomial's 1.5e-7 x |u|
    got = out[:, :N].double()
    assert bool(((got - ref).abs() <= bound).all()), float(((got - ref).abs() / bound).max())
    assert bool((out[:, N:] == 7.0).all())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(300001, 384, 96), (100003, 512, 128), (8192, 384, 96), (33333, 320, 64),
                                   (138001, 768, 192), (40000, 576, 192), (30003, 320, 192), (8200, 768, 160)])      # K > 128: two column slices
def test_mlp_dact_recompute(dt, M, N, K):
    """dU = (dY W2) * GELU'(x W1^T + b1) with the pre-activation recomputed in the kernel, + its column sums (d b1) ADDED to what is there."""
    from idvs.morec_amd import _lib, ops
    assert ops.mlp_dact_recompute_supported(M, N, K, dt) == (K <= 128)      # K = 192: the call works, the engine keeps act' there (slower)
    g = torch.Generator(device="cpu").manual_seed(M + N + K + 1)
    dy = (torch.randn(M, K, generator=g) * 0.5).to(DEV).to(dt)
    w2t = (torch.randn(N, K, generator=g) * 0.3).to(DEV).to(dt)
    x = (torch.randn(M, K, generator=g) * 0.5).to(DEV).to(dt)
    w1 = (torch.randn(N, K, generator=g) * 0.3).to(DEV).to(dt)
    b1 = torch.randn(N, generator=g).to(DEV)
    cs0 = torch.randn(N, generator=g).to(DEV)
    cs = cs0.clone()
    du = ops.mlp_dact_recompute(dy, w2t, x, w1, b1, colsum_out=cs)
    ref = (dy.double() @ w2t.double().t()) * _dgelu64(x.double() @ w1.double().t() + b1.double())
    ulp = 2.0 ** (-8 if dt == torch.bfloat16 else -11)
    bound = ulp * ref.abs() + 2e-5 * np.sqrt(K) + 1e-30     # fp32 accumulation of both products, the second one inside GELU' (|GELU''| < 0.8)
    assert bool(((du.double() - ref).abs() <= bound).all()), float(((du.double() - ref).abs() / bound).max())
    want = cs0.double() + ref.sum(0)
    tol = 1e-5 * ref.abs().sum(0) + 1e-3
    assert bool(((cs.double() - want).abs() <= tol).all()), float(((cs.double() - want).abs() / tol).max())
    # without the column sums; and the same numbers as the two-launch path (act' stored in 16 bits there: one more rounding)
    du2 = ops.mlp_dact_recompute(dy, w2t, x, w1, b1)
    assert torch.equal(du, du2)
    pre = torch.empty((M, N), device=DEV, dtype=dt)
    ops.gemm_nt(x, w1, bias=b1, act=ops.ACT_GELU, aux_out=pre, aux_deriv=True)
    old = ops.gemm_nt(dy, w2t, dact=_lib.DACT_MUL, dact_in=pre)
    assert float((old.double() - du.double()).abs().max()) <= 3 * ulp * float(ref.abs().max())


def test_mlp_dact_recompute_shape_rules():
    from idvs.morec_amd import _lib, ops
    bf = torch.bfloat16
    assert ops.mlp_dact_recompute_supported(2207744, 384, 96, bf) and ops.mlp_dact_recompute_supported(1103872, 512, 128, torch.float16)
    assert not ops.mlp_dact_recompute_supported(551936, 768, 192, bf)          # stage 2 of Swin-T: accepted by the call (two column slices) but slower than reading act'
    assert not ops.mlp_dact_recompute_supported(275968, 1024, 256, bf)         # stage 2 of Swin-B: the fragments of two [1024, 256] weights do not fit the registers
    assert not ops.mlp_dact_recompute_supported(137984, 1536, 384, bf)
    assert not ops.mlp_dact_recompute_supported(4096, 384, 96, bf) and not ops.mlp_dact_recompute_supported(300000, 384, 96, torch.float32)
    L = _lib.lib()
    assert L.morec_tuning_set(b"gemm_skinny", 1) == 0
    try:
        assert not ops.mlp_dact_recompute_supported(2207744, 384, 96, bf)
    finally:
        L.morec_tuning_set(b"gemm_skinny", 0)
    t = torch.zeros(9000, 256, device=DEV, dtype=bf)
    w = torch.zeros(1024, 256, device=DEV, dtype=bf)
    with pytest.raises(RuntimeError):
        ops.mlp_dact_recompute(t, w, t, w, None)
