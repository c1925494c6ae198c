"""-m gpu: ``morec_layernorm_fwd_res32`` / ``morec_layernorm_bwd_res32`` -- LayerNorm of the reference autocast's data flow
(``T/run.py:242``: under ``torch.cuda.amp.autocast()`` nn.Linear returns 16-bit tensors, LayerNorm takes the fp32 residual stream and
returns fp32) -- against a plain PyTorch fp32 reference of the same op: ``LayerNorm(dropout(x16 + bias) + res32 (+ pos))`` and its autograd
backward, at the row widths the towers use (BERT 768, SASRec 512, the test models' 64 / 128, a two-vector row of 1024)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref_forward(x16, bias, res, pos, period, gamma, beta, eps, keep_in=None, p_in=0.0):
    v = x16.float()
    if bias is not None:
        v = v + bias
    if keep_in is not None:
        v = v * keep_in / (1.0 - p_in)
    if res is not None:
        v = v + res
    if pos is not None:
        M = v.shape[0]
        v = v + pos[torch.arange(M, device=v.device) % period]
    z = v
    y = torch.nn.functional.layer_norm(z, (z.shape[1],), gamma, beta, eps)
    return z, y


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(300, 768), (257, 512), (100, 64), (70, 128), (64, 1024), (33, 200)])
def test_forward_and_backward_match_fp32_reference(M, N, dt):
    from idvs.morec_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 1000 + N)
    x16 = (torch.randn(M, N, generator=g) * 0.7).to(dt).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    gamma = (1.0 + 0.2 * torch.randn(N, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(N, generator=g)).to(DEV)
    eps = 1e-12
    y16, y32, z32, mean, rstd = ops.layernorm_fwd_res32(x16, gamma, beta, eps, bias=bias, res=res)
    z_ref, y_ref = _ref_forward(x16, bias, res, None, 0, gamma, beta, eps)
    assert torch.equal(z32, z_ref) or float((z32 - z_ref).abs().max()) < 1e-6
    assert float((y32 - y_ref).abs().max()) < 2e-5 * max(1.0, float(y_ref.abs().max()))
    assert torch.equal(y16, y32.to(dt))                       # the 16-bit copy is the rounded fp32 output, bit for bit
    assert float((mean - z_ref.mean(1)).abs().max()) < 1e-5 and float((rstd * torch.sqrt(z_ref.var(1, unbiased=False) + eps) - 1).abs().max()) < 1e-4
    # backward: gradient through the 16-bit copy (dy16) + gradient along the fp32 stream (dy32)
    dy16 = (torch.randn(M, N, generator=g) * 0.3).to(dt).to(DEV)
    dy32 = (torch.randn(M, N, generator=g) * 0.3).to(DEV)
    dgamma, dbeta, dbias = (torch.zeros(N, device=DEV) for _ in range(3))
    dz32, dzd16 = ops.layernorm_bwd(dy16, dy32, z32, mean, rstd, gamma, dgamma, dbeta, dbias=dbias)
    zr = z_ref.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(zr, (N,), gr, br, eps)
    yr.backward(dy16.float() + dy32)
    scale = float(zr.grad.abs().max())
    assert dz32.dtype == torch.float32 and dzd16.dtype == dt
    assert float((dz32 - zr.grad).abs().max()) < 3e-5 * max(1.0, scale)
    assert torch.equal(dzd16, dz32.to(dt))
    assert float((dgamma - gr.grad).abs().max()) < 2e-4 * max(1.0, float(gr.grad.abs().max()))
    assert float((dbeta - br.grad).abs().max()) < 2e-4 * max(1.0, float(br.grad.abs().max()))
    assert float((dbias - dzd16.float().sum(0)).abs().max()) < 2e-4 * max(1.0, float(dzd16.float().sum(0).abs().max()))
    # one-sided inputs: only the residual-stream gradient / only the GEMM gradient; no 16-bit output wanted (embedding stages)
    dz_b, none16 = ops.layernorm_bwd_res32(None, dy32, z32, mean, rstd, gamma, None, None, dt, sub16=False)
    zr.grad = None
    torch.nn.functional.layer_norm(zr, (N,), gamma, beta, eps).backward(dy32)
    assert none16 is None and float((dz_b - zr.grad).abs().max()) < 3e-5 * max(1.0, float(zr.grad.abs().max()))


def test_position_rows_and_dropout_streams():
    """The SASRec input stage (x16 + position rows -> LN -> dropout on the OUTPUT, ``T/model/modules.py:93-94``) and a sub-layer stage
    with dropout on the sub-layer output before the residual add (``modules.py:16,62``), with the masks the library exports."""
    from idvs.morec_amd import ops
    M, N, S, dt = 6 * 20, 512, 20, torch.float16
    g = torch.Generator(device="cpu").manual_seed(5)
    x16 = torch.randn(M, N, generator=g).to(dt).to(DEV)
    pos = torch.randn(S, N, generator=g).to(DEV)
    gamma, beta = (1.0 + 0.1 * torch.randn(N, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    p, seed = 0.1, 0x1234567
    y16, y32, z32, mean, rstd = ops.layernorm_fwd_res32(x16, gamma, beta, 1e-6, pos=pos, pos_period=S, p_out=p, seed_out=seed)
    keep = ops.dropout_keep_mask(M * N, p, seed).view(M, N).float()
    z_ref, y_ref = _ref_forward(x16, None, None, pos, S, gamma, beta, 1e-6)
    thr = int(p * 65536)
    inv = 1.0 / (1.0 - thr / 65536.0)
    assert float((z32 - z_ref).abs().max()) < 1e-6
    assert float((y32 - y_ref * keep * inv).abs().max()) < 3e-5 * float(y_ref.abs().max())
    assert torch.equal(y16, y32.to(dt))
    dy16 = torch.randn(M, N, generator=g).to(dt).to(DEV)
    dz, _ = ops.layernorm_bwd(dy16, None, z32, mean, rstd, gamma, None, None, p_out=p, seed_out=seed, sub16=False)
    zr = z_ref.clone().requires_grad_(True)
    (torch.nn.functional.layer_norm(zr, (N,), gamma, beta, 1e-6) * keep * inv).backward(dy16.float())
    assert float((dz - zr.grad).abs().max()) < 3e-5 * max(1.0, float(zr.grad.abs().max()))
    # sub-layer dropout (p_in): forward z = res + drop(x + bias); backward dzd16 = round(drop'(dz32)), dbias over the dropped gradient
    res = torch.randn(M, N, generator=g).to(DEV)
    bias = (0.1 * torch.randn(N, generator=g)).to(DEV)
    seed2 = 0xABCDEF01
    y16, y32, z32, mean, rstd = ops.layernorm_fwd_res32(x16, gamma, beta, 1e-6, bias=bias, res=res, p_in=p, seed_in=seed2)
    keep2 = ops.dropout_keep_mask(M * N, p, seed2).view(M, N).float()
    z_ref = (x16.float() + bias) * keep2 * inv + res
    assert float((z32 - z_ref).abs().max()) < 2e-6 * float(z_ref.abs().max())
    dbias = torch.zeros(N, device=DEV)
    dz32, dzd16 = ops.layernorm_bwd(dy16, None, z32, mean, rstd, gamma, None, None, p_in=p, seed_in=seed2, dbias=dbias)
    assert float((dzd16.float() - (dz32 * keep2 * inv).to(dt).float()).abs().max()) == 0.0
    assert float((dbias - dzd16.float().sum(0)).abs().max()) < 2e-4 * max(1.0, float(dzd16.float().sum(0).abs().max()))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(300, 768), (257, 512), (100, 64), (64, 1024)])
def test_residual_in_pre_layernorm_form_equals_the_written_stream(M, N, dt):
    """``morec_layernorm_fwd_res32_pre``: a chain of two LayerNorms where the stream between them is never written (``ops.PreLN``: the first
    call's z32 / mean / rstd / gamma / beta, recomputed by the second) gives what the chain over the written fp32 stream gives, with a
    dropout on the second sub-layer's output; row subsets (the [CLS]-only last layer) too."""
    from idvs.morec_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + 7 * N)
    xa = (torch.randn(M, N, generator=g) * 0.7).to(dt).to(DEV)
    xb = (torch.randn(M, N, generator=g) * 0.7).to(dt).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    ba, bb = ((torch.randn(N, generator=g) * 0.1).to(DEV) for _ in range(2))
    ga, gb = ((1.0 + 0.2 * torch.randn(N, generator=g)).to(DEV) for _ in range(2))
    be_a, be_b = ((0.1 * torch.randn(N, generator=g)).to(DEV) for _ in range(2))
    eps, p, seed = 1e-12, 0.1, 0xabcdef
    # written stream
    y16a, y32a, z32a, ma, ra = ops.layernorm_fwd_res32(xa, ga, be_a, eps, bias=ba, res=res)
    y16b, y32b, z32b, mb, rb = ops.layernorm_fwd_res32(xb, gb, be_b, eps, bias=bb, res=y32a, p_in=p, seed_in=seed)
    # the same chain, stream in pre-LayerNorm form
    l16a, lazy_a, lz32a, lma, lra = ops.layernorm_fwd_res32(xa, ga, be_a, eps, bias=ba, res=res, lazy_out=True)
    assert isinstance(lazy_a, ops.PreLN) and torch.equal(l16a, y16a) and torch.equal(lz32a, z32a) and torch.equal(lma, ma) and torch.equal(lra, ra)
    assert float((lazy_a.materialize() - y32a).abs().max()) < 2e-6 * max(1.0, float(y32a.abs().max()))
    l16b, lazy_b, lz32b, lmb, lrb = ops.layernorm_fwd_res32(xb, gb, be_b, eps, bias=bb, res=lazy_a, p_in=p, seed_in=seed, lazy_out=True)
    assert isinstance(lazy_b, ops.PreLN)
    tol = 2e-6 * max(1.0, float(z32b.abs().max()))
    assert float((lz32b - z32b).abs().max()) < tol                      # (fp32 contraction of the recomputed stream value at most)
    assert float((lmb - mb).abs().max()) < tol and float((lrb / rb - 1).abs().max()) < 1e-5
    assert float((l16b.float() - y16b.float()).abs().max()) <= float(y16b.float().abs().max()) * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10)
    assert float((l16b != y16b).float().mean()) < 2e-3                  # a last-bit difference may cross a 16-bit rounding boundary
    # row subset: every third row of the stream
    idx = torch.arange(0, M, 3, device=DEV, dtype=torch.int32)
    sub = lazy_a.rows(idx=idx)
    s16, _, sz, _, _ = ops.layernorm_fwd_res32(xb[idx.long()].contiguous(), gb, be_b, eps, bias=bb, res=sub, lazy_out=True)
    e16, _, ez, _, _ = ops.layernorm_fwd_res32(xb[idx.long()].contiguous(), gb, be_b, eps, bias=bb, res=y32a[idx.long()].contiguous())
    assert float((sz - ez).abs().max()) < tol
    with pytest.raises(ValueError):
        ops.layernorm_fwd_res32(xb, gb, be_b, eps, bias=bb, res=lazy_a)      # a PreLN residual cannot feed a call that has to write the stream
