"""-m gpu: ABSOLUTE-tolerance parity of the kernels that carry the benchmarked steps -- ``gemm8p_kernel`` (every epilogue mode) and
``gemm_tn8p_kernel`` -- AT THE SHAPES ``bench.py`` RUNS THEM ON: the BERT-base step (M = 54 919 real tokens of a batch / 51 200,
N x K of the QKV / output / FFN projections and their input-gradient products) and the Swin-T step (a ragged slice of stage 1's
2.2 M rows; K = 96 partial K-tile; N = 96 / 192 / 288 / 384 / 576 / 1536).  The reference is the exact product of the SAME
bf16-rounded operands in fp64 followed by the epilogue in fp64 (the arithmetic ``include/morec_hip.h: morec_gemm_nt`` states, i.e.
``torch.nn.functional.linear`` + GELU / ReLU of ``T/model/modules.py:14-17,56-63`` and HF ``BertIntermediate``).

Tolerance, elementwise: ``|out - ref| <= rtol |ref| + atol`` with
  rtol = 2^-8 for a bf16 output (one rounding to 8 significant bits), 0 for an fp32 output;
  atol = 1e-4 sigma, sigma = the standard deviation of an output element (fp32 accumulation over K, approximation error of the
         erf / exp in the activation epilogues).
A dropped K-tile, a wrong tail column or a mis-addressed row is an error of order sigma: 10^4 times the bound.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from idvs.morec_amd import _lib, ops  # noqa: E402
from idvs.morec_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, DACT_MUL  # noqa: E402
from idvs.morec_amd.engine import _splitk  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
SCALE = 0.5          # operand standard deviation: an output element has sigma = SCALE^2 sqrt(K)

M_TEXT, M_TEXT2, M_SWIN = 54919, 51200, 300001      # real tokens of a bench batch (not a multiple of 256) / padded-S layout / ragged stage-1 slice

TEXT_CASES = [
    # (M, N, K, kind)
    (M_TEXT, 2304, 768, "bias"),           # QKV projection
    (M_TEXT, 768, 768, "bias"),            # attention output projection (bias folded into the LayerNorm launch in the step; here: mode 0 + bias)
    (M_TEXT, 768, 768, "plain"),           # d(ctx) = dz W_o
    (M_TEXT, 3072, 768, "gelu_deriv"),     # FFN up + GELU, second output = GELU'(pre)
    (M_TEXT, 3072, 768, "gelu_pre"),       # ... second output = pre-activation
    (M_TEXT, 3072, 768, "dactmul_cs"),     # d(u) = (dz W_2) * act' + column sums (d b_1)
    (M_TEXT, 3072, 768, "dgelu_cs"),
    (M_TEXT, 3072, 768, "drelu_cs"),
    (M_TEXT, 3072, 768, "dactmul"),
    (M_TEXT, 768, 3072, "plain"),          # FFN down, d(x1) = du W_1
    (M_TEXT, 768, 2304, "plain"),          # d(x0) = dqkv W_qkv
    (M_TEXT, 768, 768, "f32out"),
    (M_TEXT, 2304, 768, "f32out"),
    (M_TEXT2, 2304, 768, "bias"),
    (M_TEXT2, 3072, 768, "gelu_deriv"),
    (M_TEXT2, 3072, 768, "relu_deriv"),
    (M_TEXT2, 3072, 768, "dgelu"),
    (M_TEXT2, 3072, 768, "drelu"),
    (M_TEXT2, 768, 3072, "plain"),
    (2560, 2048, 512, "relu_deriv"),       # SASRec FFN at D = 512 (forced onto the eight-phase kernel)
]
SWIN_CASES = [
    (M_SWIN, 288, 96, "bias"),             # stage-1 q|k|v (K = 96: one full + one partial K-tile)
    (M_SWIN, 96, 96, "plain"),             # stage-1 output projection (N = 96: forced)
    (M_SWIN, 384, 96, "gelu_deriv"),       # stage-1 MLP up
    (M_SWIN, 384, 96, "dactmul_cs"),
    (M_SWIN, 96, 384, "plain"),            # stage-1 MLP down
    (M_SWIN, 192, 96, "plain"),
    (M_SWIN // 4, 192, 384, "plain"),      # patch merging 4C -> 2C
    (M_SWIN // 4, 576, 192, "bias"),       # stage-2 q|k|v
    (M_SWIN // 4, 768, 192, "gelu_deriv"),
    (M_SWIN // 4, 768, 192, "dactmul_cs"),
    (M_SWIN // 4, 192, 768, "plain"),
    (M_SWIN // 16, 1152, 384, "bias"),     # stage-3 q|k|v
    (M_SWIN // 16, 1536, 384, "gelu_deriv"),
    (M_SWIN // 16, 1536, 384, "dactmul_cs"),
    (M_SWIN // 16, 384, 1536, "plain"),
]


def _operands(M, N, K, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    a = (torch.randn(M, K, device=DEV, generator=g) * SCALE).to(BF)
    b = (torch.randn(N, K, device=DEV, generator=g) * SCALE).to(BF)
    return a, b, g


def _gelu_deriv(u):
    return 0.5 * (1 + torch.erf(u / math.sqrt(2.0))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)


def _check(out, ref, rtol, atol, what):
    err = (out.double() - ref).abs()
    bound = rtol * ref.abs() + atol
    bad = err > bound
    if bool(bad.any()):
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements outside |err| <= {rtol:.3g} |ref| + {atol:.3g}; first at {idx}: "
                             f"got {float(out[tuple(idx)]):.6g} want {float(ref[tuple(idx)]):.6g}; max err {float(err.max()):.3g}")


@pytest.fixture(autouse=True)
def _force_eight_phase_kernel():
    L = _lib.lib()
    assert L.morec_tuning_set(b"gemm8p", 2) == 0      # every eligible problem on gemm8p / gemm_tn8p, whatever its size
    yield
    L.morec_tuning_set(b"gemm8p", 0)


@pytest.mark.parametrize("M,N,K,kind", TEXT_CASES + SWIN_CASES)
def test_gemm8p_absolute(M, N, K, kind):
    a, b, g = _operands(M, N, K, 7 * M + 3 * N + K)
    sigma = SCALE * SCALE * math.sqrt(K)
    acc = a.double() @ b.double().t()
    out_dt = torch.float32 if kind == "f32out" else BF
    rtol, atol = (0.0 if out_dt == torch.float32 else 2.0 ** -8), 1e-4 * sigma
    out = torch.full((M, N), 7.0, device=DEV, dtype=out_dt)
    bias = aux = din = cs = None
    kw = {}
    if kind in ("bias", "gelu_deriv", "gelu_pre", "relu_deriv"):
        bias = torch.randn(N, device=DEV, generator=g)
        acc = acc + bias.double()
    if kind.startswith(("gelu", "relu")):
        aux = torch.full((M, N), 7.0, device=DEV, dtype=BF)
        kw = dict(act=ACT_GELU if kind.startswith("gelu") else ACT_RELU, aux_out=aux, aux_deriv=kind.endswith("_deriv"))
    if kind.startswith("d"):
        din = (torch.randn(M, N, device=DEV, generator=g)).to(BF)
        kw = dict(dact={"dgelu": ACT_GELU, "drelu": ACT_RELU, "dactmul": DACT_MUL}[kind.replace("_cs", "")], dact_in=din)
        if kind.endswith("_cs"):
            cs = torch.full((N,), 0.25, device=DEV, dtype=torch.float32)
            kw["colsum_out"] = cs
    ops.gemm_nt(a, b, out=out, bias=bias, **kw)
    if kind.startswith("gelu"):
        want = torch.nn.functional.gelu(acc)
        want_aux = _gelu_deriv(acc) if kind.endswith("_deriv") else acc
    elif kind.startswith("relu"):
        want = torch.relu(acc)
        want_aux = (acc > 0).double()
    elif kind.startswith("dgelu"):
        want = acc * _gelu_deriv(din.double())
    elif kind.startswith("drelu"):
        want = acc * (din.double() > 0)
    elif kind.startswith("dactmul"):
        want = acc * din.double()
    else:
        want = acc
    _check(out, want, rtol, atol * (1.0 if not kind.startswith("dactmul") else 4.0), f"{kind} {M}x{N}x{K} out")      # |act'| operand ~ N(0, 1): up to 4 sigma
    if aux is not None:
        if kind == "relu_deriv":      # 0 / 1: exact except where the pre-activation is within rounding of zero
            assert float(((aux.double() - want_aux).abs() > 0).double().mean()) < 1e-4
        else:
            _check(aux, want_aux, rtol, atol, f"{kind} {M}x{N}x{K} aux")
    if cs is not None:      # += column sums of the output AS STORED (fp32 partial rows per 128-row block, folded in a fixed order)
        want_cs = 0.25 + out.double().sum(0)
        s_sum = math.sqrt(M) * sigma * (4.0 if kind.startswith("dactmul") else 1.0)
        assert float((cs.double() - want_cs).abs().max()) <= 1e-4 * s_sum, (float((cs.double() - want_cs).abs().max()), s_sum)


TN_CASES = [
    # (M, N, K): out[N, K] = dy[M, N]^T x[M, K]
    (M_TEXT, 2304, 768), (M_TEXT, 768, 768), (M_TEXT, 3072, 768), (M_TEXT, 768, 3072), (M_TEXT2, 3072, 768),
    (2688, 768, 512),                                       # projection head fc
    (2560, 2688, 512),                                      # scoring backward: dE = dl^T P
    (2560, 1536, 512), (2560, 512, 2048), (2560, 2048, 512),  # SASRec
    (M_SWIN, 96, 96), (M_SWIN, 288, 96), (M_SWIN, 384, 96), (M_SWIN, 96, 384),
    (M_SWIN // 4, 576, 192), (M_SWIN // 4, 768, 192), (M_SWIN // 4, 192, 768), (M_SWIN // 4, 192, 384),
    (M_SWIN // 16, 1536, 384), (M_SWIN // 16, 384, 1536),
]


@pytest.mark.parametrize("M,N,K", TN_CASES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_tn8p_absolute(M, N, K, accumulate):
    g = torch.Generator(device=DEV).manual_seed(5 * M + N + 11 * K)
    dy = (torch.randn(M, N, device=DEV, generator=g) * SCALE).to(BF)
    x = (torch.randn(M, K, device=DEV, generator=g) * SCALE).to(BF)
    base = torch.randn(N, K, device=DEV, generator=g) if accumulate else torch.full((N, K), 7.0, device=DEV)
    out = base.clone()
    # accumulate (what the engines do: += into the gradient arena) and overwrite, both with the split over tokens the engines pick for
    # this shape (the slab fold overwrites or adds; only the workspace-less atomic path needs accumulate != 0: include/morec_hip.h)
    split = _splitk(N, K, M)
    ops.gemm_tn_(dy, x, out, split_m=split, accumulate=accumulate)
    want = dy.double().t() @ x.double() + (base.double() if accumulate else 0.0)
    sigma = SCALE * SCALE * math.sqrt(M)
    _check(out, want, 0.0, 1e-4 * sigma, f"tn {M}x{N}x{K}")
    out2 = base.clone()
    ops.gemm_tn_(dy, x, out2, split_m=split, accumulate=accumulate)
    assert torch.equal(out, out2), "slab fold is not deterministic"
    if not accumulate and split > 1:      # overwrite without a workspace would race between the token chunks: refused
        with pytest.raises(RuntimeError):
            ops.gemm_tn_(dy, x, out2, split_m=split, accumulate=False, slabs=False)


def test_gemm8p_tail_split_absolute():
    """The opt-in K split of the tail round against the exact product (not only against the unsplit launch)."""
    L = _lib.lib()
    M, N, K = 51200, 768, 3072
    a, b, _ = _operands(M, N, K, 99)
    exact = a.double() @ b.double().t()
    sigma = SCALE * SCALE * math.sqrt(K)
    try:
        assert L.morec_tuning_set(b"gemm8p_tail_split", 1) == 0
        o = ops.gemm_nt(a, b)
    finally:
        L.morec_tuning_set(b"gemm8p_tail_split", 0)
    _check(o, exact, 2.0 ** -8, 1e-4 * sigma, "tail split")


@pytest.mark.parametrize("M,N,K,kind", [(M_TEXT, 768, 768, "plain"), (M_TEXT, 2304, 768, "bias"), (M_TEXT2, 768, 3072, "plain"),
                                        (M_TEXT, 3072, 768, "gelu_deriv"), (M_TEXT, 3072, 768, "dactmul_cs"), (1000, 768, 768, "plain"),
                                        (M_TEXT2 + 193, 768, 2304, "plain")])
def test_gemm8p_tile_height_is_bit_identical(M, N, K, kind):
    """Tile height (gemm8p.hip, pick_tmr): 224- / 192-row tiles (wave row 1 owns three / two 32-row blocks) change the number of rounds a
    launch takes, never a number -- outputs, second outputs and fused column sums equal the 256-row launch bit for bit; the automatic
    choice is one of the three."""
    L = _lib.lib()
    a, b, g = _operands(M, N, K, M + N + K)
    bias = torch.randn(N, device=DEV, generator=g) if kind in ("bias", "gelu_deriv") else None
    din = torch.randn(M, N, device=DEV, generator=g).to(BF) if kind == "dactmul_cs" else None

    def run():
        out = torch.full((M, N), 7.0, device=DEV, dtype=BF)
        aux = torch.full((M, N), 7.0, device=DEV, dtype=BF) if kind == "gelu_deriv" else None
        cs = torch.full((N,), 0.25, device=DEV) if kind == "dactmul_cs" else None
        kw = {}
        if kind == "gelu_deriv":
            kw = dict(act=ACT_GELU, aux_out=aux, aux_deriv=True)
        if kind == "dactmul_cs":
            kw = dict(dact=DACT_MUL, dact_in=din, colsum_out=cs)
        ops.gemm_nt(a, b, out=out, bias=bias, **kw)
        return out, aux, cs

    res = {}
    try:
        for mode in (0, 224, 192, 1):
            assert L.morec_tuning_set(b"gemm8p_tmr", mode) == 0
            res[mode] = run()
    finally:
        L.morec_tuning_set(b"gemm8p_tmr", 1)
    for mode in (224, 192, 1):
        assert torch.equal(res[mode][0], res[0][0]), f"tile height {mode}: output differs from the 256-row launch"
        if res[0][1] is not None:
            assert torch.equal(res[mode][1], res[0][1]), f"tile height {mode}: second output differs"
        if res[0][2] is not None:      # column sums: the partial rows are cut at other row boundaries, the fold order follows -- rounding only
            assert float((res[mode][2] - res[0][2]).abs().max()) <= 2e-5 * math.sqrt(M) * SCALE * SCALE * math.sqrt(K) * 4
    acc = a.double() @ b.double().t() + (bias.double() if bias is not None else 0.0)
    if kind in ("plain", "bias"):
        _check(res[224][0], acc, 2.0 ** -8, 1e-4 * SCALE * SCALE * math.sqrt(K), f"tile height 224 {M}x{N}x{K}")
