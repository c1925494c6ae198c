"""-m gpu: the latency-class NT GEMM (``csrc/gemm_small.hip``: 64 x 64 tiles on a four-stage LDS-DMA ring -- the SASRec layers' Linear products
over B S = 2 560 rows, ``T/model/modules.py:8-9,41-44``) against the 128 x 128 two-buffer kernel it replaces on the same operands: the same MFMA
and the same K order per output element, so every output -- product, bias, ReLU / GELU with the act' second output, x act' with the fused
column sums -- must be BIT-identical; and against the fp64 product of the 16-bit operands.  Ragged M / N / K (partial tiles, a K tail that is
staged through registers), strided operands, the automatic rule's shapes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(fn):
    """fn() under the 128 x 128 kernel (gemm_small off) and under gemm_small forced on every eligible shape."""
    from idvs.morec_amd import _lib
    L = _lib.lib()
    outs = []
    try:
        for mode in (1, 2):
            assert L.morec_tuning_set(b"gemm_small", mode) == 0
            outs.append(fn())
    finally:
        L.morec_tuning_set(b"gemm_small", 0)
    return outs


def _operands(M, N, K, dt, seed=0, lda=None, ldb=None):
    g = torch.Generator(device="cpu").manual_seed(seed + 3 * M + 5 * N + 7 * K)
    a = (torch.randn(M, lda or K, generator=g) * 0.5).to(DEV).to(dt)
    b = (torch.randn(N, ldb or K, generator=g) * 0.5).to(DEV).to(dt)
    return a, b


def _check_fp64(got, ref, K, dt):
    ulp = 2.0 ** (-8 if dt == torch.bfloat16 else -11)
    bound = ulp * ref.abs() + 3e-6 * np.sqrt(K) + 1e-30
    assert bool(((got.double() - ref).abs() <= bound).all()), float(((got.double() - ref).abs() / bound).max())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(2560, 512, 2048), (2560, 512, 1536), (2560, 512, 512), (2560, 1536, 512), (2688, 512, 2560),
                                   (2500, 520, 584), (100, 72, 1000), (64, 64, 64), (1, 256, 1024), (333, 264, 200), (640, 2048, 8192)])
def test_plain_product_bit_identical(dt, M, N, K):
    from idvs.morec_amd import ops
    a, b = _operands(M, N, K, dt)
    o1, o2 = _both(lambda: ops.gemm_nt(a, b).clone())
    assert torch.equal(o1, o2)
    _check_fp64(o2, a.double() @ b.double().t(), K, dt)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(2560, 2048, 512), (777, 520, 328)])
def test_epilogues_bit_identical(dt, M, N, K):
    """bias + ReLU with act'(pre) as second output (the FFN's first Linear), bias + GELU, x act' with the fused column sums (d(b1))."""
    from idvs.morec_amd import ops
    from idvs.morec_amd._lib import ACT_GELU, ACT_RELU, DACT_MUL
    a, b = _operands(M, N, K, dt, seed=1)
    bias = torch.randn(N, device=DEV)

    def relu():
        aux = torch.empty(M, N, device=DEV, dtype=dt)
        o = ops.gemm_nt(a, b, bias=bias, act=ACT_RELU, aux_out=aux, aux_deriv=True)
        return o.clone(), aux.clone()
    (o1, u1), (o2, u2) = _both(relu)
    assert torch.equal(o1, o2) and torch.equal(u1, u2)
    pre = a.double() @ b.double().t() + bias.double()
    _check_fp64(o2, torch.relu(pre), K, dt)
    # (act' = 1 where the fp32 pre-activation is positive: compare away from the zero crossing)
    far = pre.abs() > 1e-2
    assert bool((u2.double()[far] == (pre > 0).double()[far]).all())

    g1, g2 = _both(lambda: ops.gemm_nt(a, b, bias=bias, act=ACT_GELU).clone())
    assert torch.equal(g1, g2)

    dact = (torch.rand(M, N, device=DEV) > 0.5).to(dt)

    def dmul():
        cs = torch.full((N,), 0.25, device=DEV)
        o = ops.gemm_nt(a, b, dact=DACT_MUL, dact_in=dact, colsum_out=cs)
        return o.clone(), cs.clone()
    (d1, c1), (d2, c2) = _both(dmul)
    assert torch.equal(d1, d2)
    want = 0.25 + d2.double().sum(0)
    # the column sums are taken from the stored (rounded) tile rows by both kernels, folded in a different block order
    assert float((c2.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    assert float((c1.double() - c2.double()).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_strided_operands_and_output(dt):
    """A / B as column blocks of wider tensors (the fused q|k|v gradient's slices), C with a pitch wider than N: nothing written past N."""
    from idvs.morec_amd import ops
    M, N, K = 2560, 512, 1536
    a, b = _operands(M, N, K, dt, seed=2, lda=K + 64, ldb=K + 8)

    def run():
        out = torch.full((M, N + 24), 7.0, device=DEV, dtype=dt)
        ops.gemm_nt(a, b, out=out, M=M, N=N, K=K, lda=K + 64, ldb=K + 8, ldc=N + 24)
        return out.clone()
    o1, o2 = _both(run)
    assert torch.equal(o1, o2) and bool((o2[:, N:] == 7.0).all())
    _check_fp64(o2[:, :N], a[:, :K].double() @ b[:, :K].double().t(), K, dt)


def test_automatic_rule_takes_the_sasrec_shapes():
    """The automatic rule is a measured one (profiles/r06_small_gemm.txt): narrow outputs with K >= 1024.  Whatever it picks, the result does not
    depend on it."""
    from idvs.morec_amd import _lib, ops
    L = _lib.lib()
    a, b = _operands(2560, 512, 2048, torch.float16, seed=3)
    L.morec_tuning_set(b"gemm_small", 0)
    o_auto = ops.gemm_nt(a, b).clone()
    o_off, o_on = _both(lambda: ops.gemm_nt(a, b).clone())
    assert torch.equal(o_auto, o_off) and torch.equal(o_auto, o_on)
