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
