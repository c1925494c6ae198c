"""Tensor-level wrappers over the C-ABI (``include/morec_hip.h``): they marshal ``torch`` device tensors
into raw pointers + sizes and launch on torch's current HIP stream.  PyTorch is plumbing here (memory,
streams); all arithmetic happens in ``libmorec_hip.so``.  Every wrapper raises on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, BF16, F16, F32, AttnDesc, CeDesc, GemmDesc, SwinAttnDesc, check

FLT_MIN_MASK = -3.4028234663852886e38  # torch.finfo(torch.float32).min: HF eager additive key mask


# Deterministic mode (``MOREC_DETERMINISTIC=1`` in the environment, or ``set_deterministic(True)``): the library's kernels that end in
# fp32 atomics (LayerNorm / bias column sums, the embedding-table scatters) fold per-block partials in a fixed order instead, and the
# engines keep the split-K of the exact-fp32 weight-gradient GEMMs at one -- two runs of the same step give the same bits.  The
# reference sets torch's deterministic flags (``T/run.py:313-314``).  Covers the text / ID towers; see DESIGN.md §3.
DETERMINISTIC = _lib.env_flag("MOREC_DETERMINISTIC")      # the library is told the same at load (_lib.lib)


def set_deterministic(on: bool = True):
    global DETERMINISTIC
    check(_lib.lib().morec_tuning_set(b"deterministic", int(bool(on))), "morec_tuning_set(deterministic)")
    DETERMINISTIC = bool(on)


def code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise _lib.MorecError(f"unsupported dtype {dt}")


def _p(t):
    """Raw device address (an int: every pointer parameter is declared ``c_void_p`` in ``_lib._SIGS``, ctypes converts; no wrapper object per argument)."""
    return None if t is None else t.data_ptr()


# the handle without building a torch.cuda.Stream object per launch (MOREC_STREAM_OBJ=1: the object path, for A/B timing of the host side)
_raw_stream = None if os.environ.get("MOREC_STREAM_OBJ") == "1" else getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's CURRENT stream on the current device as a ``hipStream_t`` (every launch goes there; ~700 calls per training step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def stream_wait_stream(waiting, signal=None):
    """``waiting`` (raw handle) runs nothing issued later before what ``signal`` (default: torch's current stream) holds now has finished."""
    check(_lib.lib().morec_stream_wait_stream(waiting, _stream() if signal is None else signal), "morec_stream_wait_stream")


def _dev(t):
    if not t.is_cuda:
        raise _lib.MorecError("libmorec_hip needs device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise _lib.MorecError("tensor must be contiguous")
    return t


def is16(dt: torch.dtype) -> bool:
    """bf16 or fp16: the two 16-bit storage types of the MFMA paths (``MOREC_BF16`` / ``MOREC_F16``)."""
    return dt in (torch.bfloat16, torch.float16)


def pad8(n: int) -> int:
    return (n + 7) & ~7


# ---------------------------------------------------------------------------------------------------------
# How products of fp32 operands run: "exact" = the exact-fp32 MFMA (the parity mode pinned to the reference goldens), "bf16x3" = both
# operands split into bf16 hi / lo parts and ONE bf16 GEMM over [hi | hi | lo] x [hi | lo | hi] (fp32 accumulation; the lo.lo term,
# 2^-16 relative, is dropped).  Set per step / forward by whoever owns the model (``compute_dtype`` "fp32" / "fp32x3").
FP32_GEMM = "exact"


class fp32_gemm_mode:
    """``with ops.fp32_gemm_mode("bf16x3"): ...`` -- sets ``FP32_GEMM`` for the block and restores it (also drops the per-step split cache
    on both ends).  ``Model.forward`` / ``TrainStep.forward_backward`` run under their model's mode; the autograd shells remember the
    mode of their forward for their backward.  Elsewhere (an encoder called on its own, eval) the process default "exact" applies."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        global FP32_GEMM
        self.prev, FP32_GEMM = FP32_GEMM, self.mode
        _X3_CACHE.clear()
        return self

    def __exit__(self, *exc):
        global FP32_GEMM
        FP32_GEMM = self.prev
        _X3_CACHE.clear()
        return False


def split_bf16x3(t, lo_slot, rows=None, cols=None, ld=None):
    """fp32 [R, C] (row pitch ``ld``) -> bf16 [R, 3 C]: hi | hi | lo (``lo_slot`` 2, A side) or hi | lo | hi (``lo_slot`` 1, B side)."""
    R = t.shape[0] if rows is None else rows
    Cc = t.shape[1] if cols is None else cols
    ld = t.stride(0) if ld is None else ld
    out = torch.empty((R, 3 * Cc), device=t.device, dtype=torch.bfloat16)
    check(_lib.lib().morec_split_bf16x3(_p(t), _p(out), R, Cc, ld, 3 * Cc, lo_slot, _stream()), "morec_split_bf16x3")
    return out


# hi | hi | lo splits of whole activation tensors, kept for the length of one step: the forward's split of x is what the weight-gradient
# product of the same x needs again, the dX product's split of dY likewise.  An entry holds a reference to its source (so the address
# cannot be recycled under it) and the source's version counter (an in-place write invalidates it).
_X3_CACHE = {}


def x3_cache_clear():
    _X3_CACHE.clear()


X3_VERIFY = os.environ.get("MOREC_X3_VERIFY", "0") == "1"
X3_HITS = 0


def split_cached(t):
    """``split_bf16x3(t, 2)`` ([hi | hi | lo], bf16 [R, 3 C]) of a whole contiguous fp32 tensor, computed once per step.
    The entry is keyed by the tensor object; ``t._version`` only sees torch's own in-place writes -- the library's kernels write through raw
    pointers -- so the contract is: no morec op overwrites a tensor between two GEMMs that read it (the engines keep it: the one in-place
    form, ``layernorm_fwd(z_inplace=True)``, overwrites a GEMM OUTPUT that was never an operand).  ``ops.X3_VERIFY`` (env ``MOREC_X3_VERIFY=1``)
    re-splits on every hit and compares: ``tests/test_fp32x3_gpu.py`` runs whole training steps under it."""
    global X3_HITS
    key = id(t)
    hit = _X3_CACHE.get(key)
    if hit is not None and hit[0] is t and hit[1] == t._version:
        if X3_VERIFY:
            X3_HITS += 1
            if not torch.equal(split_bf16x3(t, 2), hit[2]):
                raise RuntimeError("split_cached: the tensor changed since its [hi | hi | lo] split was cached (stale operand)")
        return hit[2]
    s = split_bf16x3(t, 2)
    _X3_CACHE[key] = (t, t._version, s)
    return s


def gemm_tn_x3_(dy3, x3, out, N, K, split_m=1, stream=None):
    """out[N, K] += dy^T x for fp32 dy [M, N], x [M, K] given as their [hi | hi | lo] splits (bf16 [M, 3 N], [M, 3 K]): the three
    products hi.hi + hi.lo + lo.hi on the transposing bf16 GEMM, reading the column blocks in place (row pitch 3 N / 3 K)."""
    _dev(dy3), _dev(x3)
    M = dy3.shape[0]
    ws = None
    if split_m > 1:
        need = _lib.lib().morec_gemm_tn_workspace_bytes(N, K, split_m) // 4
        ws = _scratch(_TN_WS, dy3.device, need, 20 * 1024 * 1024)
    py, px = dy3.data_ptr(), x3.data_ptr()
    for sy, sx in ((0, 0), (0, 2), (2, 0)):      # (hi, hi), (hi, lo), (lo, hi)
        check(_lib.lib().morec_gemm_tn(C.c_void_p(py + 2 * sy * N), C.c_void_p(px + 2 * sx * K), _p(out), M, N, K, 3 * N, 3 * K, out.stride(0),
                                       BF16, split_m, 1, _p(ws), _stream() if stream is None else stream), "morec_gemm_tn")
    return out


_CS_WS = {}      # per-device fp32 scratch of the fused column sums (stream-ordered reuse, like _TN_WS)
# Scratch buffers that have been outgrown are KEPT (never handed back to the allocator): a captured graph of the step
# (TrainStep.step_graphed) has their addresses baked into its kernel arguments and keeps writing there at every replay.
_WS_RETIRED = []


def _scratch(cache, device, need, floor):
    ws = cache.get(device)
    if ws is None or ws.numel() < need:
        grown = 0
        if ws is not None:
            _WS_RETIRED.append(ws)
            grown = ws.numel() + ws.numel() // 2      # geometric growth: everything ever retired stays below twice the live buffer
        ws = torch.empty(max(need, floor, grown), device=device, dtype=torch.float32)
        cache[device] = ws
    return ws


_GEMM_DESCS = {}


def gemm_nt(a, b, *, bias=None, act=ACT_NONE, out=None, out_dtype=None, aux_out=None, dact=ACT_NONE, dact_in=None,
            accumulate=0, split_k=1, alpha=1.0, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, colsum_out=None,
            aux_deriv=False):
    """out[M,N] (+)= alpha * a[M,K] @ b[N,K]^T with the fused epilogues of ``morec_gemm_nt``; ``colsum_out`` (fp32 [N],
    ``dact`` epilogues only) additionally receives ``+= out.sum(0)``: the bias gradient of the layer below."""
    _dev(a), _dev(b)
    M = a.shape[0] if M is None else M
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=out_dtype or a.dtype)
    ldc = out.stride(0) if ldc is None else ldc
    if FP32_GEMM == "bf16x3" and a.dtype == torch.float32 and b.dtype == torch.float32 and K % 8 == 0:
        whole = a.dim() == 2 and a.is_contiguous() and M == a.shape[0] and K == a.shape[1] and lda == K
        a = split_cached(a) if whole else split_bf16x3(a, 2, M, K, lda)
        b = split_bf16x3(b, 1, N, K, ldb)
        K, lda, ldb = 3 * K, 3 * K, 3 * K
    key = (M, N, K, lda, ldb, ldc, a.dtype, out.dtype, act, dact, accumulate, split_k, alpha, bool(aux_deriv))
    d = _GEMM_DESCS.get(key)
    if d is None:      # the descriptor is read during the call only: one object per distinct launch shape, reused while the cache holds it
        if len(_GEMM_DESCS) >= 8192:      # the unpadded token layout makes M a per-batch number: bound the cache instead of growing with the run
            _GEMM_DESCS.clear()
        d = _GEMM_DESCS[key] = GemmDesc(M, N, K, lda, ldb, ldc, code(a.dtype), code(out.dtype), act, dact, accumulate, split_k, alpha, int(bool(aux_deriv)))
    if colsum_out is not None:
        need = _lib.lib().morec_gemm_colsum_workspace_bytes(M, N) // 4
        ws = _scratch(_CS_WS, a.device, need, 4 * 1024 * 1024)
        check(_lib.lib().morec_gemm_nt_colsum(C.byref(d), _p(a), _p(b), _p(out), _p(bias), _p(aux_out), _p(dact_in),
                                              _p(colsum_out), _p(ws), _stream()), "morec_gemm_nt_colsum")
        return out
    check(_lib.lib().morec_gemm_nt(C.byref(d), _p(a), _p(b), _p(out), _p(bias), _p(aux_out), _p(dact_in), _stream()),
          "morec_gemm_nt")
    return out


_DR_WS = {}


def mlp_dact_recompute_supported(M, N, K, dtype):
    """Does ``mlp_dact_recompute`` take this shape (Swin stage-1 / stage-2 MLP: 288 < N <= 768, K <= 192, M >= 8192, 16-bit)?  The forward then
    runs fc1 + GELU WITHOUT the act'(pre) output."""
    return dtype in (torch.bfloat16, torch.float16) and bool(_lib.lib().morec_mlp_dact_recompute_supported(M, N, K, code(dtype)))


def mlp_dact_recompute(dy, w2t, x, w1, b1, colsum_out=None):
    """dU = (dy @ w2t^T) * GELU'(x @ w1^T + b1), ``colsum_out += dU.sum(0)``: the backward of ``GELU(fc1(x)) -> fc2`` down to the
    pre-activation gradient, with the pre-activation recomputed from ``x`` instead of read from a saved tensor
    (``morec_mlp_dact_recompute``)."""
    _dev(dy), _dev(x)
    M, K = dy.shape
    N = w2t.shape[0]
    assert dy.is_contiguous() and x.is_contiguous() and w2t.is_contiguous() and w1.is_contiguous()
    assert x.shape == (M, K) and w1.shape == (N, K) and w2t.shape == (N, K)
    out = torch.empty((M, N), device=dy.device, dtype=dy.dtype)
    ws = None
    if colsum_out is not None:
        ws = _scratch(_DR_WS, dy.device, _lib.lib().morec_mlp_dact_recompute_workspace_bytes(N) // 4, 1024 * 1024)
    check(_lib.lib().morec_mlp_dact_recompute(_p(dy), _p(w2t), _p(x), _p(w1), _p(b1), _p(out), _p(colsum_out), _p(ws), M, N, K,
                                              code(dy.dtype), _stream()), "morec_mlp_dact_recompute")
    return out


_TN_WS = {}


def gemm_tn_(dy, x, out, split_m=1, accumulate=True, slabs=True, stream=None):
    """out[N, K] (+)= dy[M, N]^T @ x[M, K] (bf16 operands, fp32 out) without transposed copies.  ``slabs``: reduce the
    split-m partials through a reusable fp32 workspace (deterministic) instead of fp32 atomics.  ``stream``: raw handle of the stream to
    launch on (default: torch's current stream) -- the weight-gradient stream, without a ``torch.cuda.stream`` context per launch."""
    _dev(dy), _dev(x)
    M, N = dy.shape
    K = x.shape[1]
    ws = None
    if slabs and split_m > 1:
        need = _lib.lib().morec_gemm_tn_workspace_bytes(N, K, split_m) // 4
        ws = _scratch(_TN_WS, dy.device, need, 20 * 1024 * 1024)
    check(_lib.lib().morec_gemm_tn(_p(dy), _p(x), _p(out), M, N, K, dy.stride(0), x.stride(0), out.stride(0), code(dy.dtype),
                                   split_m, int(accumulate), _p(ws), _stream() if stream is None else stream), "morec_gemm_tn")
    return out


def transpose(x, out=None, out_dtype=None, ld_out=None):
    """[R, C] -> [C, ld_out>=R] (pad columns, if any, are zero)."""
    _dev(x)
    R, Cc = x.shape
    ld_out = pad8(R) if ld_out is None else ld_out
    if out is None:
        alloc = torch.zeros if ld_out != R else torch.empty
        out = alloc((Cc, ld_out), device=x.device, dtype=out_dtype or x.dtype)
    check(_lib.lib().morec_transpose(_p(x), _p(out), R, Cc, x.stride(0), out.stride(0), code(x.dtype), code(out.dtype),
                                     _stream()), "morec_transpose")
    return out


def cast(x, out_dtype, out=None):
    _dev(x)
    if out is None:
        out = torch.empty_like(x, dtype=out_dtype)
    check(_lib.lib().morec_cast(_p(x), _p(out), x.numel(), code(x.dtype), code(out.dtype), _stream()), "morec_cast")
    return out


def act_bwd(dy, pre, act):
    out = torch.empty_like(dy)
    check(_lib.lib().morec_act_bwd(_p(_dev(dy)), _p(_dev(pre)), _p(out), dy.numel(), act, code(dy.dtype), _stream()),
          "morec_act_bwd")
    return out


def scaled_sum(xs, scale, out=None):
    """``scale * (xs[0] + xs[1] + xs[2])`` with fp32 arithmetic and one rounding (``morec_scaled_sum``; one to three same-shaped tensors):
    the mean over the text attributes of an item (``T/model/encoders.py:113-116``) and, with one input, the gradient each attribute's
    encoder pass receives."""
    xs = [_dev(x) for x in xs]
    assert 1 <= len(xs) <= 3 and all(x.shape == xs[0].shape and x.dtype == xs[0].dtype for x in xs)
    if out is None:
        out = torch.empty_like(xs[0])
    ptr = [_p(x) for x in xs] + [None] * (3 - len(xs))
    check(_lib.lib().morec_scaled_sum(ptr[0], ptr[1], ptr[2], _p(out), xs[0].numel(), float(scale), code(xs[0].dtype), _stream()), "morec_scaled_sum")
    return out


def colsum_(x, out, M=None, N=None, ld=None):
    """out[N] += column sums of x[M, N] (fp32 atomics)."""
    _dev(x)
    M = x.shape[0] if M is None else M
    N = x.shape[1] if N is None else N
    ld = x.stride(0) if ld is None else ld
    check(_lib.lib().morec_colsum(_p(x), _p(out), M, N, ld, code(x.dtype), _stream()), "morec_colsum")
    return out


def layernorm_fwd(x, gamma, beta, eps, *, bias=None, res=None, pos=None, pos_period=0, save_z=True, z_inplace=False,
                  p_in=0.0, seed_in=0, p_out=0.0, seed_out=0, rowscale=None, rows_per_scale=0):
    _dev(x)
    M, N = x.shape
    y = torch.empty_like(x)
    need_z = save_z and (bias is not None or res is not None or pos is not None or p_in > 0 or rowscale is not None)
    z = (x if z_inplace else torch.empty_like(x)) if need_z else None
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    check(_lib.lib().morec_layernorm_fwd(_p(x), _p(bias), _p(res), _p(pos), pos_period, _p(gamma), _p(beta), eps, _p(z),
                                         _p(y), _p(mean), _p(rstd), M, N, code(x.dtype), p_in, seed_in, p_out, seed_out,
                                         _p(rowscale), rows_per_scale, _stream()), "morec_layernorm_fwd")
    return y, (z if need_z else x), mean, rstd


class PreLN:
    """The fp32 residual stream in PRE-LayerNorm form: the stream value is ``LN(z) = (z - mean[m]) * rstd[m] * gamma + beta`` of tensors a
    LayerNorm call has saved anyway; the consumer (the next ``layernorm_fwd_res32``) recomputes it in registers, so the stream itself is never
    written or read back (``morec_layernorm_fwd_res32_pre``)."""
    __slots__ = ("z", "mean", "rstd", "gamma", "beta")

    def __init__(self, z, mean, rstd, gamma, beta):
        self.z, self.mean, self.rstd, self.gamma, self.beta = z, mean, rstd, gamma, beta

    def rows(self, idx=None, stride=None, n=None):
        """The same stream restricted to rows ``idx`` (int tensor) or ``0, stride, 2 stride, ...`` (``n`` rows)."""
        if idx is not None:
            i = idx.long()
            return PreLN(self.z.index_select(0, i), self.mean.index_select(0, i), self.rstd.index_select(0, i), self.gamma, self.beta)
        N = self.z.shape[1]
        return PreLN(self.z.view(n, stride, N)[:, 0].contiguous(), self.mean.view(n, stride)[:, 0].contiguous(),
                     self.rstd.view(n, stride)[:, 0].contiguous(), self.gamma, self.beta)

    def materialize(self):
        return ((self.z - self.mean[:, None]) * self.rstd[:, None]) * self.gamma[None, :] + self.beta[None, :]


def layernorm_fwd_res32(x16, gamma, beta, eps, *, bias=None, res=None, pos=None, pos_period=0, p_in=0.0, seed_in=0, p_out=0.0, seed_out=0,
                        lazy_out=False):
    """LayerNorm of the autocast data flow (``morec_layernorm_fwd_res32``; ``*_res32`` compute modes): ``x16`` is the 16-bit output of the
    sub-layer's GEMM, ``res`` the fp32 residual stream (a tensor, or a ``PreLN``).  Returns (y16, y32, z32, mean, rstd): the next GEMM's
    operand, the fp32 residual stream, and what the backward needs.  ``lazy_out``: the stream is NOT written; ``y32`` is a ``PreLN`` over
    (z32, mean, rstd, gamma, beta) for the next call to recompute (needs ``p_out == 0``)."""
    _dev(x16)
    M, N = x16.shape
    z = torch.empty((M, N), device=x16.device, dtype=torch.float32)
    y16 = torch.empty_like(x16)
    mean = torch.empty(M, device=x16.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x16.device, dtype=torch.float32)
    if lazy_out and p_out > 0:
        raise ValueError("layernorm_fwd_res32(lazy_out=True) cannot carry an output dropout")
    if isinstance(res, PreLN):
        if not lazy_out or pos is not None or p_out > 0:
            raise ValueError("a PreLN residual needs lazy_out=True, no pos, no output dropout")
        check(_lib.lib().morec_layernorm_fwd_res32_pre(_p(x16), _p(bias), _p(res.z), _p(res.mean), _p(res.rstd), _p(res.gamma), _p(res.beta),
                                                       _p(gamma), _p(beta), eps, _p(z), _p(y16), _p(mean), _p(rstd), M, N, code(x16.dtype),
                                                       p_in, seed_in, _stream()), "morec_layernorm_fwd_res32_pre")
        return y16, PreLN(z, mean, rstd, gamma, beta), z, mean, rstd
    y32 = None if lazy_out else torch.empty((M, N), device=x16.device, dtype=torch.float32)
    check(_lib.lib().morec_layernorm_fwd_res32(_p(x16), _p(bias), _p(res), _p(pos), pos_period, _p(gamma), _p(beta), eps, _p(z), _p(y32), _p(y16),
                                               _p(mean), _p(rstd), M, N, code(x16.dtype), p_in, seed_in, p_out, seed_out, _stream()),
          "morec_layernorm_fwd_res32")
    return y16, (PreLN(z, mean, rstd, gamma, beta) if lazy_out else y32), z, mean, rstd


def layernorm_bwd_res32(dy16, dy32, z, mean, rstd, gamma, dgamma, dbeta, dtype16, p_in=0.0, seed_in=0, p_out=0.0, seed_out=0, dbias=None, sub16=True):
    """Backward of ``layernorm_fwd_res32``: returns (dz32, dzd16) -- the fp32 gradient along the residual stream and the 16-bit gradient of
    the sub-layer output (None with ``sub16=False``: embedding stages, where nothing 16-bit sits below the LayerNorm)."""
    M, N = z.shape
    dz = torch.empty((M, N), device=z.device, dtype=torch.float32)
    dzd = torch.empty((M, N), device=z.device, dtype=dtype16) if sub16 else None
    check(_lib.lib().morec_layernorm_bwd_res32(_p(dy16), _p(dy32), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dz), _p(dzd), _p(dgamma), _p(dbeta),
                                               _p(dbias), M, N, code(dtype16), p_in, seed_in, p_out, seed_out, _stream()), "morec_layernorm_bwd_res32")
    return dz, dzd


def layernorm_bwd(dy_a, dy_b, z, mean, rstd, gamma, dgamma, dbeta, p_in=0.0, seed_in=0, p_out=0.0, seed_out=0, dbias=None,
                  dres=None, rowscale=None, rows_per_scale=0, sub16=True):
    """Returns (dz, dzd): dz feeds the residual branch, dzd = dropout / DropPath backward of dz feeds the sub-layer (dzd is
    dz when p_in = 0 and there is no rowscale).  ``dres``: gradient arriving at z along a pre-LN residual stream.
    An fp32 ``z`` with a 16-bit ``dy_a`` is the autocast data flow (``layernorm_fwd_res32``): dz is then the fp32 residual-stream gradient
    (``dy_b`` is fp32 too) and dzd the 16-bit sub-layer gradient (``sub16=False``: not wanted, None)."""
    _dev(dy_a)
    if z.dtype == torch.float32 and is16(dy_a.dtype):
        assert dres is None and rowscale is None and (dy_b is None or dy_b.dtype == torch.float32)
        return layernorm_bwd_res32(dy_a, dy_b, z, mean, rstd, gamma, dgamma, dbeta, dy_a.dtype, p_in, seed_in, p_out, seed_out, dbias, sub16)
    M, N = z.shape
    dz = torch.empty_like(z)
    dzd = torch.empty_like(z) if (p_in > 0 or rowscale is not None) else None
    check(_lib.lib().morec_layernorm_bwd(_p(dy_a), _p(dy_b), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dz), _p(dzd),
                                         _p(dgamma), _p(dbeta), _p(dbias), M, N, code(z.dtype), p_in, seed_in, p_out, seed_out,
                                         _p(dres), _p(rowscale), rows_per_scale, _stream()), "morec_layernorm_bwd")
    return dz, (dz if dzd is None else dzd)


class TransposeBatch:
    """dst_i = src_i^T for a fixed list of (src [R, C], dst [C, ld >= R]) pairs in ONE launch (``morec_transpose_batch``).  The
    tensors must stay where they are (the device table holds their addresses): persistent weight shadows and their W^T copies."""

    @staticmethod
    def eligible(src, dst):
        R, C_ = src.shape
        return (src.dim() == 2 and src.is_contiguous() and dst.stride(1) == 1 and src.dtype == dst.dtype and C_ % 4 == 0
                and dst.stride(0) % 4 == 0 and dst.stride(0) >= ((R + 3) & ~3) and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0)

    def __init__(self, pairs):
        assert pairs, "empty transpose batch"
        items = (_lib.TransposeItem * len(pairs))()
        tile0 = 0
        for i, (src, dst) in enumerate(pairs):
            assert TransposeBatch.eligible(src, dst), "morec_transpose_batch: shape / alignment rules (include/morec_hip.h)"
            R, C_ = src.shape
            items[i] = _lib.TransposeItem(src.data_ptr(), dst.data_ptr(), R, C_, C_, dst.stride(0), tile0, 0)
            tile0 += ((R + 63) // 64) * ((C_ + 63) // 64)
        self.n_items, self.n_tiles, self.dtype = len(pairs), tile0, pairs[0][0].dtype
        self.keep = pairs
        host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        self.table = host.to(pairs[0][0].device)

    def run(self):
        check(_lib.lib().morec_transpose_batch(_p(self.table), self.n_items, self.n_tiles, code(self.dtype), _stream()), "morec_transpose_batch")


def pos_grad_(dz, dpos, period):
    M, N = dz.shape
    check(_lib.lib().morec_pos_grad(_p(dz), _p(dpos), M, N, period, code(dz.dtype), _stream()), "morec_pos_grad")


# Upper bound of the SPARE rows a packed token layout may carry behind its last sequence (token counts padded up to a multiple of this by
# ``engine.token_packing_host(pad_to=...)`` so that one captured graph serves every batch of the bucket): the attention launches append
# ceil(SPARE_ROWS_MAX / 16) blocks that zero those rows of their outputs.
SPARE_ROWS_MAX = 1024


def attn_desc(n_seq, T, n_heads, dh, causal, scale, mask_value, dtype, p_drop=0.0, seed=0, cu_seqlens=None, total_rows=0):
    """``cu_seqlens``: int32 [n_seq + 1] device tensor for the unpadded token layout (the caller keeps it alive); ``total_rows``: rows of
    the packed buffers (spare rows behind the last sequence are zero-filled in ctx / dqkv)."""
    return AttnDesc(n_seq, T, n_heads, dh, int(causal), scale, mask_value, code(dtype), p_drop, seed,
                    None if cu_seqlens is None else cu_seqlens.data_ptr(), int(total_rows) if cu_seqlens is not None else 0,
                    SPARE_ROWS_MAX if (cu_seqlens is not None and total_rows) else 0)


def attn_fwd(desc, qkv, key_keep):
    _dev(qkv), _dev(key_keep)
    ctx = torch.empty((qkv.shape[0], qkv.shape[1] // 3), device=qkv.device, dtype=qkv.dtype)
    check(_lib.lib().morec_attn_fwd(C.byref(desc), _p(qkv), _p(key_keep), _p(ctx), _stream()), "morec_attn_fwd")
    return ctx


def attn_bwd(desc, qkv, key_keep, dctx, dbias=None):
    """``dbias``: fp32 [3 H] gradient buffer of the fused q|k|v bias, accumulated into (column sums of dqkv)."""
    _dev(dctx)
    dqkv = torch.empty_like(qkv)
    if dbias is None:
        check(_lib.lib().morec_attn_bwd(C.byref(desc), _p(qkv), _p(key_keep), _p(dctx), _p(dqkv), _stream()), "morec_attn_bwd")
    else:
        ws = torch.empty((desc.n_seq, qkv.shape[1]), device=qkv.device, dtype=torch.float32)
        check(_lib.lib().morec_attn_bwd_dbias(C.byref(desc), _p(qkv), _p(key_keep), _p(dctx), _p(dqkv), qkv.shape[0], _p(dbias),
                                              _p(ws), ws.numel() * 4, _stream()), "morec_attn_bwd_dbias")
    return dqkv


def bert_embed_fwd(ids32, word, pos, type0, gamma, beta, eps, T, dtype, p_out=0.0, seed_out=0):
    M = ids32.numel()
    H = word.shape[1]
    z = torch.empty((M, H), device=word.device, dtype=dtype)
    y = torch.empty_like(z)
    mean = torch.empty(M, device=word.device, dtype=torch.float32)
    rstd = torch.empty(M, device=word.device, dtype=torch.float32)
    check(_lib.lib().morec_bert_embed_fwd(_p(ids32), _p(word), _p(pos), _p(type0), _p(gamma), _p(beta), eps, _p(z),
                                          _p(y), _p(mean), _p(rstd), M, T, H, code(dtype), p_out, seed_out, _stream()),
          "morec_bert_embed_fwd")
    return y, z, mean, rstd


def bert_embed_bwd_(ids32, dz, dword, dpos, dtype0, pad_id, T, order=None):
    """``order``: int32 argsort of ``ids32`` (integer bookkeeping done by the caller) -- enables the run-length scatter."""
    M, H = dz.shape
    check(_lib.lib().morec_bert_embed_bwd(_p(ids32), _p(dz), _p(dword), _p(dpos), _p(dtype0), pad_id, M, T, H,
                                          code(dz.dtype), _p(order), _stream()), "morec_bert_embed_bwd")


def gather_rows(table, idx32, dtype):
    R, D = idx32.numel(), table.shape[1]
    out = torch.empty((R, D), device=table.device, dtype=dtype)
    check(_lib.lib().morec_gather_rows(_p(table), _p(idx32), _p(out), R, D, code(dtype), _stream()), "morec_gather_rows")
    return out


def scatter_add_rows_(d, idx32, dtable, pad_id):
    R, D = d.shape
    check(_lib.lib().morec_scatter_add_rows(_p(_dev(d)), _p(idx32), _p(dtable), R, D, pad_id, code(d.dtype), _stream()),
          "morec_scatter_add_rows")


def indexed_rows_copy(src, out, in_idx=None, out_idx=None, R=None):
    """out[out_idx[r]] = src[in_idx[r]] (int32 device index vectors; None = identity)."""
    _dev(src), _dev(out)
    R = (in_idx if in_idx is not None else out_idx).numel() if R is None else R
    check(_lib.lib().morec_indexed_rows_copy(_p(src), _p(out), _p(in_idx), _p(out_idx), R, src.shape[1], code(src.dtype), _stream()),
          "morec_indexed_rows_copy")
    return out


def strided_rows_copy(src, out, R, D, in_stride, out_stride):
    check(_lib.lib().morec_strided_rows_copy(_p(src), _p(out), R, D, in_stride, out_stride, code(src.dtype), _stream()),
          "morec_strided_rows_copy")
    return out


# ---------------------------------------------------------------------------------------------------------
def ce_desc(B, S, D, Nc, col_offset, dtype, dE_fp32=False, ws_from_fwd=False):
    """``ws_from_fwd`` (backward): the workspace is the untouched one of the matching forward call -- its tables are reused."""
    return CeDesc(B, S, D, Nc, col_offset, code(dtype), int(bool(dE_fp32)), int(bool(ws_from_fwd)))


def ce_workspace(desc, device):
    n = _lib.lib().morec_inbatch_ce_workspace_bytes(C.byref(desc))
    return torch.empty(n, device=device, dtype=torch.uint8)


def inbatch_ce_fwd(desc, P, E, row_ids, col_ids, col_logpop, col_valid, row_valid, ws):
    Nr = desc.B * desc.S
    lse = torch.empty(Nr, device=P.device, dtype=torch.float32)
    row_loss = torch.empty(Nr, device=P.device, dtype=torch.float32)
    loss_sum = torch.zeros(1, device=P.device, dtype=torch.float32)
    check(_lib.lib().morec_inbatch_ce_fwd(C.byref(desc), _p(_dev(P)), _p(_dev(E)), _p(row_ids), _p(col_ids), _p(col_logpop),
                                          _p(col_valid), _p(row_valid), _p(lse), _p(row_loss), _p(loss_sum), _p(ws),
                                          _stream()), "morec_inbatch_ce_fwd")
    return loss_sum, lse, row_loss


def inbatch_ce_bwd(desc, P, E, row_ids, col_ids, col_logpop, col_valid, row_valid, lse, gscale_dev, gscale, ws):
    """Returns (dP, dE); dE is fp32 when the descriptor asks for it (``dE_fp32``: the pooled step reduces it over ranks in fp32)."""
    dP = torch.empty_like(P)
    dE = torch.empty_like(E, dtype=torch.float32) if desc.dE_fp32 else torch.empty_like(E)
    check(_lib.lib().morec_inbatch_ce_bwd(C.byref(desc), _p(P), _p(E), _p(row_ids), _p(col_ids), _p(col_logpop),
                                          _p(col_valid), _p(row_valid), _p(lse), _p(gscale_dev), gscale, _p(dP), _p(dE),
                                          _p(ws), _stream()), "morec_inbatch_ce_bwd")
    return dP, dE


def bce_fwd(P, E, row_valid, B, S):
    """BCE variant scoring (``morec_bce_fwd``): returns (loss_sum fp32[1], scores fp32[2, B*S])."""
    _dev(P), _dev(E)
    scores = torch.empty((2, B * S), device=P.device, dtype=torch.float32)
    loss_sum = torch.zeros(1, device=P.device, dtype=torch.float32)
    check(_lib.lib().morec_bce_fwd(_p(P), _p(E), _p(row_valid), _p(scores), _p(loss_sum), B, S, P.shape[-1], code(P.dtype), _stream()),
          "morec_bce_fwd")
    return loss_sum, scores


def bce_bwd(P, E, row_valid, scores, gscale_dev, B, S):
    dP, dE = torch.empty_like(P), torch.empty_like(E)
    check(_lib.lib().morec_bce_bwd(_p(P), _p(E), _p(row_valid), _p(scores), _p(gscale_dev), _p(dP), _p(dE), B, S, P.shape[-1],
                                   code(P.dtype), _stream()), "morec_bce_bwd")
    return dP, dE


def adamw_(param, grad, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    check(_lib.lib().morec_adamw(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(shadow), param.numel(), lr, beta1,
                                 beta2, eps, wd, step, grad_scale, _stream()), "morec_adamw")


_SEED_SOURCE_OWNER = None      # id() of the StepParams whose seed word the library currently reads (process-wide, like the knob itself)


class StepParams:
    """The device-resident step block (``morec_step_params``: step count, bias corrections, loss scale, overflow flag): what the
    reference keeps in ``GradScaler()`` + AdamW's ``step`` (``T/run.py:210,243-247``).  ``init_scale`` 1.0 = no loss scaling."""

    def __init__(self, device, init_scale=1.0, step=0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, dynamic=True):
        self.buf = torch.zeros(64, device=device, dtype=torch.uint8)
        self.f32 = self.buf.view(torch.float32)       # [16]: loss_scale at index 4
        self.i32 = self.buf.view(torch.int32)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.dynamic = bool(dynamic)
        check(_lib.lib().morec_step_params_init(_p(self.buf), float(init_scale), int(step), _stream()), "morec_step_params_init")

    @property
    def loss_scale_dev(self):
        """fp32 [1] view of the scale the next backward pass multiplies the loss gradient with (no host read)."""
        return self.f32[4:5]

    def use_as_seed_source(self, on: bool = True):
        """Register (or clear) this block's ``drop_seed_mixed`` word as the library's dropout seed source (``morec_dropout_seed_source``):
        every ``decide_`` then changes the masks of the following launches, whatever their seed arguments -- what a replayed graph needs."""
        global _SEED_SOURCE_OWNER
        if not on and _SEED_SOURCE_OWNER not in (None, id(self)):
            return                                   # another block is registered: leave it alone
        ptr = C.c_void_p(self.buf.data_ptr() + _lib.StepParams.drop_seed_mixed.offset) if on else None
        check(_lib.lib().morec_dropout_seed_source(ptr), "morec_dropout_seed_source")
        _SEED_SOURCE_OWNER = id(self) if on else None

    def __del__(self):      # the library must not keep a pointer into a block that is going away
        try:
            if _SEED_SOURCE_OWNER == id(self):
                self.use_as_seed_source(False)
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def check_finite_(self, grad):
        check(_lib.lib().morec_grad_check_finite(_p(grad), grad.numel(), _p(self.buf), _stream()), "morec_grad_check_finite")

    def decide_(self, beta1, beta2):
        check(_lib.lib().morec_step_decide(_p(self.buf), beta1, beta2, self.growth_factor, self.backoff_factor, self.growth_interval,
                                           int(self.dynamic), _stream()), "morec_step_decide")

    def host(self):
        """Host copy of the block (synchronises): for logging / checkpoints, never inside the step."""
        raw = bytes(self.buf.cpu().numpy().tobytes())
        return _lib.StepParams.from_buffer_copy(raw)


def adamw_sp_(param, grad, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, wd, sp: StepParams):
    """``morec_adamw_sp``: AdamW with step count / bias corrections / 1 / loss-scale read from the device block; a no-op when the
    step was marked for skipping (non-finite gradient)."""
    check(_lib.lib().morec_adamw_sp(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(shadow), code(shadow.dtype) if shadow is not None else BF16,
                                    param.numel(), lr, beta1, beta2, eps, wd, _p(sp.buf), _stream()), "morec_adamw_sp")


def eval_rank(prec, item_emb, hist32, target32):
    U, D = prec.shape
    rank = torch.empty(U, device=prec.device, dtype=torch.int32)
    ts = torch.empty(U, device=prec.device, dtype=torch.float32)
    check(_lib.lib().morec_eval_rank(_p(_dev(prec)), _p(_dev(item_emb)), _p(_dev(hist32)), hist32.shape[1], _p(target32),
                                     _p(rank), _p(ts), U, item_emb.shape[0], D, _stream()), "morec_eval_rank")
    return rank


# ---------------------------------------------------------------------------------------------------------
# Swin vision tower
# ---------------------------------------------------------------------------------------------------------
def swin_attn_desc(n_img, H, W, window, shift, heads, dh, dtype):
    return SwinAttnDesc(n_img, H, W, window, shift, heads, dh, float(dh) ** -0.5, code(dtype))


def swin_attn_fwd(desc, qkv, bias_t):
    _dev(qkv), _dev(bias_t)
    ctx = torch.empty((qkv.shape[0], qkv.shape[1] // 3), device=qkv.device, dtype=qkv.dtype)
    check(_lib.lib().morec_swin_attn_fwd(C.byref(desc), _p(qkv), _p(bias_t), _p(ctx), _stream()), "morec_swin_attn_fwd")
    return ctx


def swin_attn_bwd(desc, qkv, bias_t, ctx, dctx, dbias_t=None, dbqkv=None):
    """``dbqkv``: fp32 [3 C] gradient buffer of the fused q|k|v bias, accumulated into (column sums of dqkv)."""
    _dev(dctx), _dev(ctx)
    dqkv = torch.empty_like(qkv)
    if dbqkv is None:
        check(_lib.lib().morec_swin_attn_bwd(C.byref(desc), _p(qkv), _p(bias_t), _p(ctx), _p(dctx), _p(dqkv), _p(dbias_t),
                                             _stream()), "morec_swin_attn_bwd")
    else:
        n_win = desc.n_img * (desc.H // desc.window) * (desc.W // desc.window)
        ws = torch.empty(((n_win + 3) // 4 * 4, qkv.shape[1]), device=qkv.device, dtype=torch.float32)     # one row per wavefront slot (4 per block)
        check(_lib.lib().morec_swin_attn_bwd_dbias(C.byref(desc), _p(qkv), _p(bias_t), _p(ctx), _p(dctx), _p(dqkv), _p(dbias_t),
                                                   _p(dbqkv), _p(ws), ws.numel() * 4, _stream()), "morec_swin_attn_bwd_dbias")
    return dqkv


def swin_bias_expand(table, window):
    heads = table.shape[1]
    out = torch.empty((heads, window ** 2, window ** 2), device=table.device, dtype=torch.float32)
    check(_lib.lib().morec_swin_bias_expand(_p(_dev(table)), _p(out), window, heads, _stream()), "morec_swin_bias_expand")
    return out


def swin_bias_reduce_(dbias_t, dtable, window):
    check(_lib.lib().morec_swin_bias_reduce(_p(_dev(dbias_t)), _p(dtable), window, dtable.shape[1], _stream()),
          "morec_swin_bias_reduce")


def swin_patchify(pixels, patch, dtype):
    _dev(pixels)
    if pixels.dtype != torch.float32:
        raise _lib.MorecError("pixels must be fp32 NCHW")
    n, c, R, _ = pixels.shape
    G = R // patch
    out = torch.empty((n * G * G, c * patch * patch), device=pixels.device, dtype=dtype)
    check(_lib.lib().morec_swin_patchify(_p(pixels), _p(out), n, c, R, patch, out.stride(0), code(dtype), _stream()),
          "morec_swin_patchify")
    return out


def swin_patchify_u8(pixels_hwc, patch, dtype, mean=0.5, std=0.5, out=None):
    """uint8 [n, R, R, 3] decoded images -> normalised patch rows (ToTensor + Normalize(0.5, 0.5), V/data_utils/dataset.py:69-73).
    ``out``: a caller-owned [n G G, 3 patch^2] buffer of ``dtype`` (the double-buffered input feed, ``data_utils.images.DeviceImageFeed``)."""
    _dev(pixels_hwc)
    if pixels_hwc.dtype != torch.uint8:
        raise _lib.MorecError("expected uint8 HWC images")
    n, R, _, c = pixels_hwc.shape
    G = R // patch
    if out is None:
        out = torch.empty((n * G * G, c * patch * patch), device=pixels_hwc.device, dtype=dtype)
    else:
        assert out.shape == (n * G * G, c * patch * patch) and out.dtype == dtype and out.is_contiguous()
    check(_lib.lib().morec_swin_patchify_u8(_p(pixels_hwc), _p(out), n, c, R, patch, out.stride(0), mean, std, code(dtype), _stream()),
          "morec_swin_patchify_u8")
    return out


def swin_merge(x, n_img, H, W, Cc, reverse=False):
    _dev(x)
    out = torch.empty((n_img * H * W, Cc) if reverse else (n_img * (H // 2) * (W // 2), 4 * Cc), device=x.device, dtype=x.dtype)
    check(_lib.lib().morec_swin_merge(_p(x), _p(out), n_img, H, W, Cc, int(reverse), code(x.dtype), _stream()), "morec_swin_merge")
    return out


def swin_pool_fwd(x, n_img, tokens):
    out = torch.empty((n_img, x.shape[1]), device=x.device, dtype=x.dtype)
    check(_lib.lib().morec_swin_pool_fwd(_p(_dev(x)), _p(out), n_img, tokens, x.shape[1], code(x.dtype), _stream()), "morec_swin_pool_fwd")
    return out


def swin_pool_bwd(dout, n_img, tokens):
    dx = torch.empty((n_img * tokens, dout.shape[1]), device=dout.device, dtype=dout.dtype)
    check(_lib.lib().morec_swin_pool_bwd(_p(_dev(dout)), _p(dx), n_img, tokens, dout.shape[1], code(dout.dtype), _stream()), "morec_swin_pool_bwd")
    return dx


def bias_residual(a, bias, res, rowscale=None, rows_per_scale=0, inplace=True):
    """res + rowscale[row / rows_per_scale] * (a + bias), written over ``a`` by default."""
    _dev(a), _dev(res)
    out = a if inplace else torch.empty_like(a)
    check(_lib.lib().morec_bias_residual(_p(a), _p(bias), _p(res), _p(rowscale), rows_per_scale, _p(out), a.shape[0], a.shape[1],
                                         code(a.dtype), _stream()), "morec_bias_residual")
    return out


def droppath_scale(n, p, seed, device="cuda"):
    out = torch.empty(n, device=device, dtype=torch.float32)
    check(_lib.lib().morec_droppath_scale(_p(out), n, p, seed, _stream()), "morec_droppath_scale")
    return out


def dropout_keep_mask(n, p, seed, device="cuda"):
    out = torch.empty(n, device=device, dtype=torch.uint8)
    check(_lib.lib().morec_dropout_keep_mask(_p(out), n, p, seed, _stream()), "morec_dropout_keep_mask")
    return out


def probe(device="cuda"):
    out = torch.zeros(4096, device=device, dtype=torch.int32)
    check(_lib.lib().morec_probe(_p(out), _stream()), "morec_probe")
    return out


def image_resize_u8_packed(flat, meta, tabs, R: int, device, out=None, src_buf=None):
    """``morec_image_resize_u8`` over a batch the HOST has already packed (``data_utils.images.pack_images``; CPU tensors, ideally
    page-locked by a collate thread): three asynchronous H2D copies + the resize kernel on the current stream -> uint8 [n, R, R, 3].
    ``out`` / ``src_buf``: caller-owned device buffers for the result and for the packed source bytes (``DeviceImageFeed``)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.MorecError("libmorec_hip needs device tensors (no CPU fallback)")
    n = meta.shape[0]
    if src_buf is not None:
        assert src_buf.numel() >= flat.numel() and src_buf.dtype == torch.uint8
        src = src_buf[:flat.numel()]
        src.copy_(flat, non_blocking=True)
    else:
        src = flat.to(device, non_blocking=True)
    meta_d = meta.to(device, non_blocking=True)
    tabs_d = tabs.to(device, non_blocking=True)
    if out is None:
        out = torch.empty((n, R, R, 3), device=device, dtype=torch.uint8)
    else:
        assert out.shape == (n, R, R, 3) and out.dtype == torch.uint8 and out.is_contiguous()
    check(_lib.lib().morec_image_resize_u8(_p(src), _p(meta_d), _p(tabs_d), _p(out), n, R, _stream()), "morec_image_resize_u8")
    return out


def image_resize_u8(images, R: int, device=None):
    """Decoded uint8 [H, W, 3] images of arbitrary sizes (numpy arrays) -> uint8 [n, R, R, 3] on the device: one pinned H2D
    copy of the packed bytes + ``morec_image_resize_u8`` (Pillow's BILINEAR resampler, bit for bit -- what the reference's
    dataloader workers do per image with ``tv.transforms.Resize``, ``V/data_utils/dataset.py:68-73``)."""
    from .data_utils.images import pack_images
    device = torch.device("cuda") if device is None else torch.device(device)
    if device.type != "cuda":
        raise _lib.MorecError("libmorec_hip needs device tensors (no CPU fallback)")
    flat, meta, tabs = pack_images(images, R)
    return image_resize_u8_packed(torch.from_numpy(flat).pin_memory(), torch.from_numpy(meta), torch.from_numpy(tabs), R, device)
