"""Host-side orchestration of the HIP kernels: explicit forward AND hand-derived backward for the three
stages of ``Model.forward`` (``T/model/model.py:31-69``) -- item encoder, SASRec user encoder, in-batch CE.

Nothing here does arithmetic in PyTorch: tensors are device buffers handed to ``ops`` (C-ABI launchers).
Both the ``torch.autograd.Function`` wrappers in ``functional.py`` (drop-in ``Model``) and the fused
``TrainStep`` driver call these functions.

Layout: activations are row-major ``[tokens, features]`` in the compute dtype (fp32 = exact-fp32 MFMA
parity mode, bf16 = fast mode); parameters are fp32 masters; every Linear weight is "prepared" once per
step into the compute dtype in both orientations (``W`` for ``Y = X W^T`` and ``W^T`` for ``dX = dY W``).
Weight gradients ``dW = dY^T X`` run as split-K NT GEMMs over transposed activations with fp32 atomics.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch

from . import ops
from ._lib import ACT_GELU, ACT_RELU, DACT_MUL


@dataclass
class LayerCfg:
    """One post-LN transformer layer (``T/model/modules.py:66-75`` / HF ``BertLayer``)."""
    H: int
    heads: int
    T: int               # tokens per sequence (S for SASRec, num_words_title for BERT)
    act: int             # ACT_RELU (SASRec FFN) | ACT_GELU (BERT)
    eps: float
    causal: bool
    mask_value: float    # additive value on masked keys: -1e9 (T/model/encoders.py:27) | finfo.min (HF eager)
    # the reference autocast's data flow (T/run.py:242: LayerNorm runs and returns fp32 under torch.cuda.amp.autocast): the residual stream
    # is fp32, GEMM operands / outputs 16-bit.  The layer's input / output is then a PAIR (x16, x32): the next GEMM's operand and the
    # residual stream (``ops.layernorm_fwd_res32``); the backward tells the two flows apart by the dtypes of what it is handed.
    res32: bool = False


_GOLD = 0x9E3779B97F4A7C15


@dataclass
class DropCfg:
    """Training-mode dropout of one encoder stack: hidden-state probability, attention-probability probability and the
    per-step seed; ``site(k)`` derives the independent stream of the k-th dropout site of the stack."""
    p_hidden: float = 0.0
    p_attn: float = 0.0
    seed: int = 0

    def site(self, k: int) -> int:
        return (self.seed + (k + 1) * _GOLD) & 0xFFFFFFFFFFFFFFFF

    def stream(self, j: int) -> "DropCfg":
        """Independent dropout streams for the j-th pass of the SAME stack within one step (one pass per text attribute,
        ``T/model/encoders.py:107-112``: every call of the shared Text_Encoder draws its own masks); j = 0 is this configuration itself."""
        if j == 0 or (self.p_hidden <= 0 and self.p_attn <= 0):
            return self
        return DropCfg(self.p_hidden, self.p_attn, (self.seed ^ (j * 0xC2B2AE3D27D4EB4F)) & 0xFFFFFFFFFFFFFFFF)


NO_DROP = DropCfg()


@dataclass
class PreparedLinear:
    w: torch.Tensor      # [out, in]  compute dtype
    wt: torch.Tensor     # [in, pad8(out)] compute dtype (zero padded)


def prepare_linear(weight: torch.Tensor, dtype: torch.dtype, shadow: torch.Tensor | None = None,
                   shadow_t: torch.Tensor | None = None) -> PreparedLinear:
    """``shadow``: an up-to-date copy of ``weight`` already in the compute dtype (the AdamW kernel's bf16 shadow);
    ``shadow_t``: its transpose, already refreshed for this step (``TrainStep``'s one batched transpose launch)."""
    if shadow is not None and shadow.dtype == dtype:
        w = shadow
    else:
        w = weight if weight.dtype == dtype else ops.cast(weight.contiguous(), dtype)
    if shadow_t is not None and shadow_t.dtype == dtype and w is shadow:
        return PreparedLinear(w, shadow_t)
    wt = ops.transpose(w.contiguous(), out_dtype=dtype)
    return PreparedLinear(w.contiguous(), wt)


def _cdiv(a: int, b: int) -> int:
    return (a + b - 1) // b


def _splitk(N: int, K: int, Mp: int) -> int:
    """Split factor over the token dimension for dW[N, K] = dY^T X so that the launch fills the 256 CUs once with
    256 x 256 tiles (1 workgroup of 8 waves per CU) or, for small weights, twice with 128 x 128 tiles."""
    big = _cdiv(N, 256) * _cdiv(K, 256)
    s = max(1, min(256 // big, Mp // 1024)) if big <= 256 else 1      # floor: 10 tiles x 26 chunks = 260 workgroups would be TWO rounds on 256 CUs
    if big * s >= 192:
        return s
    small = _cdiv(N, 128) * _cdiv(K, 128)
    return max(1, min(round(512 / small), max(1, Mp // 512), 64))


class WgradStream:
    """Weight gradients on a SECOND HIP stream.  dW = dY^T X feeds nothing in the backward pass (only the gradient reduce / AdamW at
    the end of the step), while the persistent 256 x 256 GEMMs of the dX chain leave the last, partly filled round of tiles with idle
    CUs (600 tiles on 256 CUs = 2.34 rounds) and the slab folds / small kernels between them leave more: launched on their own stream
    the dW workgroups start on every CU the dX chain does not use at that moment, and vice versa -- the hardware dispatcher fills
    both kernels' tails with the other's tiles.  Ordering: the side stream waits for the producing kernels (event); the operands of
    its launches are kept ALIVE (a reference list) until ``join()`` -- end of the backward pass, before anything reads the gradient
    arenas -- has made the main stream wait for the side stream: whatever is freed after that point is reused by main-stream
    allocations that are ordered behind the wait.  (Not ``Tensor.record_stream``: it defers the reuse of ~20 GB of activations per
    step until the side stream's events have COMPLETED on the device, and a host that issues steps faster than the GPU runs them
    -- there is no synchronisation inside a step -- then allocates fresh memory for every step in flight until the caching allocator
    has to stall and flush: 34 -> 140 ms/step, intermittently.)  ``MOREC_WGRAD_STREAM=0`` keeps everything on one stream."""
    enabled = os.environ.get("MOREC_WGRAD_STREAM", "1") != "0"
    _streams: dict = {}
    _dirty: set = set()
    _keep: dict = {}

    @classmethod
    def get(cls, device):
        if not cls.enabled or device.type != "cuda":
            return None
        st = cls._streams.get(device)
        if st is None:
            st = cls._streams[device] = torch.cuda.Stream(device=device)
        return st

    @classmethod
    def join(cls, device):
        """Main stream waits for every weight-gradient launch issued so far (no-op when nothing is pending)."""
        if device in cls._dirty:
            torch.cuda.current_stream(device).wait_stream(cls._streams[device])
            cls._dirty.discard(device)
        cls._keep.pop(device, None)      # the operands of the side stream's launches may be freed (and reused behind the wait) now


def linear_wgrad_(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, dyt=None, xt=None):
    """dw[N, K] += dy[M, N]^T @ x[M, K]  (split over M, slabs folded in a fixed order; on the weight-gradient stream, see ``WgradStream``).
    The accumulation into ``dw`` is a plain read-modify-write (``morec_gemm_tn``: single writer): every call that adds into the same
    ``dw`` goes through THIS function, i.e. onto the one weight-gradient stream (or the current stream when there is none), in issue order."""
    if ops.is16(dy.dtype) and dyt is None and xt is None and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0:
        M, N, K = dy.shape[0], dy.shape[1], x.shape[1]
        side = WgradStream.get(dy.device)
        if side is None:
            ops.gemm_tn_(dy, x, dw, split_m=_splitk(N, K, M))    # transpose-free: ds_read_b64_tr_b16 fragments
            return None, None
        # dy (and the zeroed arena) are produced on the main stream.  Raw handles, one foreign call for the ordering, none for a stream context:
        # 17 ... 57 of these per step, and on the launch-bound configurations the host path is the step time (scripts/host_profile.py:
        # wait_stream + `with torch.cuda.stream(side)` were 0.95 of BERT-tiny's 1.9 ms)
        sh = side.cuda_stream
        ops.stream_wait_stream(sh)
        ops.gemm_tn_(dy, x, dw, split_m=_splitk(N, K, M), stream=sh)
        WgradStream._keep.setdefault(dy.device, []).extend((dy, x))
        WgradStream._dirty.add(dy.device)
        return None, None
    if (dy.dtype == torch.float32 and ops.FP32_GEMM == "bf16x3" and dyt is None and xt is None and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0
            and dy.is_contiguous() and x.is_contiguous()):
        # fp32x3 mode: the [hi | hi | lo] splits the dX product / the forward product of the same tensors have already made, three
        # transposing bf16 GEMMs (no transposed copies), on the weight-gradient stream like the bf16 path
        M, N, K = dy.shape[0], dy.shape[1], x.shape[1]
        dy3, x3 = ops.split_cached(dy), ops.split_cached(x)
        side = WgradStream.get(dy.device)
        if side is None:
            ops.gemm_tn_x3_(dy3, x3, dw, N, K, split_m=_splitk(N, K, M))
            return None, None
        sh = side.cuda_stream
        ops.stream_wait_stream(sh)
        ops.gemm_tn_x3_(dy3, x3, dw, N, K, split_m=_splitk(N, K, M), stream=sh)
        WgradStream._keep.setdefault(dy.device, []).extend((dy3, x3))
        WgradStream._dirty.add(dy.device)
        return None, None
    dyt = ops.transpose(dy) if dyt is None else dyt
    xt = ops.transpose(x) if xt is None else xt
    N, K, Mp = dyt.shape[0], xt.shape[0], dyt.shape[1]
    # (deterministic mode: one K range per output element -- a single contribution per launch instead of split-K atomics in arrival order)
    ops.gemm_nt(dyt, xt, out=dw, accumulate=2, split_k=1 if ops.DETERMINISTIC else _splitk(N, K, Mp))
    return dyt, xt


# ---------------------------------------------------------------------------------------------------------
# one transformer layer
# ---------------------------------------------------------------------------------------------------------
# res32 modes: the fp32 residual stream between the LayerNorms of the encoder layers is handed on as ``ops.PreLN`` (the previous LayerNorm's
# saved input + statistics) and recomputed by its one reader instead of being written and read back: the forward LayerNorm moves 12 instead
# of 16 bytes per element.  MOREC_RES32_LAZY=0 keeps the written stream (A/B, tests).
RES32_LAZY = os.environ.get("MOREC_RES32_LAZY", "1") != "0"


def layer_forward(cfg: LayerCfg, w: dict, x0: torch.Tensor, key_keep: torch.Tensor, n_seq: int, need_grad: bool,
                  drop: DropCfg = NO_DROP, site0: int = 0, cu=None):
    """w keys: qkv (PreparedLinear [3H,H]), bqkv, o, bo, ln1_g, ln1_b, f1, b1, f2, b2, ln2_g, ln2_b.
    Dropout sites of a layer: site0 = attention probabilities, site0+1 = attention sub-layer output, site0+2 = FFN output.
    The FFN's first GEMM leaves the activation's DERIVATIVE act'(x1 W1^T + b1) beside its output (one erf / exp evaluation serves
    both), so the backward's dU = (dZ W2) * act' is a plain multiply in that GEMM's epilogue."""
    dh = cfg.H // cfg.heads
    x0r = None
    if cfg.res32:      # (x16, x32): the GEMM operand and the fp32 residual stream
        x0, x0r = x0
    desc = ops.attn_desc(n_seq, cfg.T, cfg.heads, dh, cfg.causal, 1.0 / math.sqrt(dh), cfg.mask_value, x0.dtype,
                         drop.p_attn, drop.site(site0), cu, total_rows=x0.shape[0])
    ph, s1, s2 = drop.p_hidden, drop.site(site0 + 1), drop.site(site0 + 2)
    qkv = ops.gemm_nt(x0, w["qkv"].w, bias=w["bqkv"])
    ctx = ops.attn_fwd(desc, qkv, key_keep)
    a = ops.gemm_nt(ctx, w["o"].w)
    if cfg.res32:
        # (the stream between the LayerNorms of the encoder layers travels as ops.PreLN: never written, recomputed by its one reader)
        x1, x1r, z1, mean1, rstd1 = ops.layernorm_fwd_res32(a, w["ln1_g"], w["ln1_b"], cfg.eps, bias=w["bo"], res=x0r, p_in=ph, seed_in=s1,
                                                            lazy_out=RES32_LAZY)
    else:
        x1, z1, mean1, rstd1 = ops.layernorm_fwd(a, w["ln1_g"], w["ln1_b"], cfg.eps, bias=w["bo"], res=x0, z_inplace=True,
                                                 p_in=ph, seed_in=s1)
    u = torch.empty((x0.shape[0], w["f1"].w.shape[0]), device=x0.device, dtype=x0.dtype) if need_grad else None
    g = ops.gemm_nt(x1, w["f1"].w, bias=w["b1"], act=cfg.act, aux_out=u, aux_deriv=need_grad)      # u = act'(pre-activation)
    f = ops.gemm_nt(g, w["f2"].w)
    if cfg.res32:
        x2, x2r, z2, mean2, rstd2 = ops.layernorm_fwd_res32(f, w["ln2_g"], w["ln2_b"], cfg.eps, bias=w["b2"], res=x1r, p_in=ph, seed_in=s2,
                                                            lazy_out=RES32_LAZY)
    else:
        x2, z2, mean2, rstd2 = ops.layernorm_fwd(f, w["ln2_g"], w["ln2_b"], cfg.eps, bias=w["b2"], res=x1, z_inplace=True,
                                                 p_in=ph, seed_in=s2)
    saved = (desc, x0, qkv, ctx, z1, mean1, rstd1, x1, u, g, z2, mean2, rstd2, key_keep, ph, s1, s2) if need_grad else None
    return ((x2, x2r) if cfg.res32 else x2), saved


def layer_backward(cfg: LayerCfg, w: dict, saved, dx2_a, dx2_b, g: dict, need_dx: bool = True):
    """g: fp32 gradient buffers (accumulated into): qkv [3H,H], bqkv, o, bo, ln1_g, ln1_b, f1, b1, f2, b2, ln2_g, ln2_b
    (bias entries may be None).  Returns (da, db) with dx0 = da + db; ``need_dx=False`` (nothing trainable below this
    layer) skips the input-gradient GEMM and returns (None, None)."""
    desc, x0, qkv, ctx, z1, mean1, rstd1, x1, u, gact, z2, mean2, rstd2, key_keep, ph, s1, s2 = saved
    # dz* = gradient at the residual sum (also the residual branch's gradient); dzd* = after the sub-layer's dropout
    dz2, dzd2 = ops.layernorm_bwd(dx2_a, dx2_b, z2, mean2, rstd2, w["ln2_g"], g["ln2_g"], g["ln2_b"], p_in=ph, seed_in=s2,
                                  dbias=g.get("b2"))
    linear_wgrad_(dzd2, gact, g["f2"])
    du = ops.gemm_nt(dzd2, w["f2"].wt, dact=DACT_MUL, dact_in=u, K=dzd2.shape[1], N=u.shape[1], colsum_out=g.get("b1"))   # x act', + d(b1)
    linear_wgrad_(du, x1, g["f1"])
    dx1 = ops.gemm_nt(du, w["f1"].wt, K=du.shape[1], N=x1.shape[1])
    dz1, dzd1 = ops.layernorm_bwd(dx1, dz2, z1, mean1, rstd1, w["ln1_g"], g["ln1_g"], g["ln1_b"], p_in=ph, seed_in=s1,
                                  dbias=g.get("bo"))
    linear_wgrad_(dzd1, ctx, g["o"])
    dctx = ops.gemm_nt(dzd1, w["o"].wt, K=dzd1.shape[1], N=ctx.shape[1])
    dqkv = ops.attn_bwd(desc, qkv, key_keep, dctx, dbias=g.get("bqkv"))
    linear_wgrad_(dqkv, x0, g["qkv"])
    if not need_dx:
        return None, None
    dx0 = ops.gemm_nt(dqkv, w["qkv"].wt, K=dqkv.shape[1], N=x0.shape[1])
    return dx0, dz1


def gather_cls(x, n_seq: int, T: int, cu=None):
    """Rows of the first token of every sequence: stride T in the padded layout, ``cu_seqlens[:-1]`` in the packed one."""
    out = torch.empty((n_seq, x.shape[1]), device=x.device, dtype=x.dtype)
    if cu is None:
        return ops.strided_rows_copy(x, out, n_seq, x.shape[1], T, 1)
    return ops.indexed_rows_copy(x, out, in_idx=cu, R=n_seq)


def scatter_cls(xc, out, n_seq: int, T: int, cu=None):
    if cu is None:
        return ops.strided_rows_copy(xc, out, n_seq, xc.shape[1], 1, T)
    return ops.indexed_rows_copy(xc, out, out_idx=cu, R=n_seq)


def layer_forward_cls(cfg: LayerCfg, w: dict, x0: torch.Tensor, key_keep: torch.Tensor, n_seq: int, need_grad: bool,
                      drop: DropCfg = NO_DROP, site0: int = 0, cu=None):
    """LAST encoder layer when only token 0 of every sequence is consumed downstream (``hidden[:, 0]``,
    ``T/model/encoders.py:69``): K and V are still needed for every token, but the attention output projection, both
    LayerNorms and the FFN are row-wise, so they run on the n_seq [CLS] rows only -- 9/12 of the layer's GEMM work is
    skipped for 29/30 of the tokens.  The reference computes (and discards) all rows; the kept rows are bit-for-bit the
    same arithmetic.  Returns x2 restricted to the [CLS] rows: [n_seq, H]."""
    dh = cfg.H // cfg.heads
    H, T = cfg.H, cfg.T
    x0r = None
    if cfg.res32:
        x0, x0r = x0
    desc = ops.attn_desc(n_seq, T, cfg.heads, dh, cfg.causal, 1.0 / math.sqrt(dh), cfg.mask_value, x0.dtype,
                         drop.p_attn, drop.site(site0), cu, total_rows=x0.shape[0])
    ph, s1, s2 = drop.p_hidden, drop.site(site0 + 1), drop.site(site0 + 2)
    qkv = ops.gemm_nt(x0, w["qkv"].w, bias=w["bqkv"])
    ctx = ops.attn_fwd(desc, qkv, key_keep)
    ctx_c = gather_cls(ctx, n_seq, T, cu)
    if cfg.res32 and isinstance(x0r, ops.PreLN):      # the residual rows of a stream that exists as the previous LayerNorm's input only
        x0_c = x0r.rows(idx=cu[:-1]) if cu is not None else x0r.rows(stride=T, n=n_seq)
    else:
        x0_c = gather_cls(x0r if cfg.res32 else x0, n_seq, T, cu)      # the residual rows (res32: from the fp32 stream)
    a = ops.gemm_nt(ctx_c, w["o"].w)
    if cfg.res32:
        x1, x1r, z1, mean1, rstd1 = ops.layernorm_fwd_res32(a, w["ln1_g"], w["ln1_b"], cfg.eps, bias=w["bo"], res=x0_c, p_in=ph, seed_in=s1,
                                                            lazy_out=RES32_LAZY)
    else:
        x1, z1, mean1, rstd1 = ops.layernorm_fwd(a, w["ln1_g"], w["ln1_b"], cfg.eps, bias=w["bo"], res=x0_c, z_inplace=True,
                                                 p_in=ph, seed_in=s1)
    u = torch.empty((n_seq, w["f1"].w.shape[0]), device=x0.device, dtype=x0.dtype) if need_grad else None
    g = ops.gemm_nt(x1, w["f1"].w, bias=w["b1"], act=cfg.act, aux_out=u, aux_deriv=need_grad)      # u = act'(pre-activation)
    f = ops.gemm_nt(g, w["f2"].w)
    if cfg.res32:      # (the [CLS] vectors feed the projection head's GEMM: only the 16-bit copy is consumed)
        x2, _, z2, mean2, rstd2 = ops.layernorm_fwd_res32(f, w["ln2_g"], w["ln2_b"], cfg.eps, bias=w["b2"], res=x1r, p_in=ph, seed_in=s2,
                                                          lazy_out=RES32_LAZY)
    else:
        x2, z2, mean2, rstd2 = ops.layernorm_fwd(f, w["ln2_g"], w["ln2_b"], cfg.eps, bias=w["b2"], res=x1, z_inplace=True,
                                                 p_in=ph, seed_in=s2)
    saved = (desc, x0, qkv, ctx_c, z1, mean1, rstd1, x1, u, g, z2, mean2, rstd2, key_keep, ph, s1, s2, n_seq, cu) if need_grad else None
    return x2, saved


def layer_backward_cls(cfg: LayerCfg, w: dict, saved, dx2_c: torch.Tensor, g: dict, need_dx: bool = True):
    """Backward of ``layer_forward_cls``: dx2_c is the gradient at the [CLS] rows [n_seq, H].  Returns (da, db) over ALL
    tokens with dx0 = da + db (db carries the residual-branch gradient, non-zero on the [CLS] rows only)."""
    desc, x0, qkv, ctx_c, z1, mean1, rstd1, x1, u, gact, z2, mean2, rstd2, key_keep, ph, s1, s2, n_seq, cu = saved
    H, T = cfg.H, cfg.T
    dz2, dzd2 = ops.layernorm_bwd(dx2_c, None, z2, mean2, rstd2, w["ln2_g"], g["ln2_g"], g["ln2_b"], p_in=ph, seed_in=s2,
                                  dbias=g.get("b2"))
    linear_wgrad_(dzd2, gact, g["f2"])
    du = ops.gemm_nt(dzd2, w["f2"].wt, dact=DACT_MUL, dact_in=u, K=dzd2.shape[1], N=u.shape[1], colsum_out=g.get("b1"))   # x act', + d(b1)
    linear_wgrad_(du, x1, g["f1"])
    dx1 = ops.gemm_nt(du, w["f1"].wt, K=du.shape[1], N=H)
    dz1, dzd1 = ops.layernorm_bwd(dx1, dz2, z1, mean1, rstd1, w["ln1_g"], g["ln1_g"], g["ln1_b"], p_in=ph, seed_in=s1,
                                  dbias=g.get("bo"))
    linear_wgrad_(dzd1, ctx_c, g["o"])
    dctx_c = ops.gemm_nt(dzd1, w["o"].wt, K=dzd1.shape[1], N=H)
    dctx = torch.zeros((x0.shape[0], H), device=dctx_c.device, dtype=dctx_c.dtype)
    scatter_cls(dctx_c, dctx, n_seq, T, cu)
    dqkv = ops.attn_bwd(desc, qkv, key_keep, dctx, dbias=g.get("bqkv"))
    linear_wgrad_(dqkv, x0, g["qkv"])
    if not need_dx:
        return None, None
    dx0 = ops.gemm_nt(dqkv, w["qkv"].wt, K=dqkv.shape[1], N=H)
    if dz1.dtype == dctx.dtype:
        dres = dctx.zero_()                  # reuse: residual-branch gradient, [CLS] rows only
    else:                                    # res32: the residual stream's gradient is fp32
        dres = torch.zeros((x0.shape[0], H), device=dz1.device, dtype=dz1.dtype)
    scatter_cls(dz1, dres, n_seq, T, cu)
    return dx0, dres


# ---------------------------------------------------------------------------------------------------------
# SASRec user encoder (T/model/encoders.py:7-28, T/model/modules.py:78-96)
# ---------------------------------------------------------------------------------------------------------
UE = "user_encoder.transformer_encoder."


def sasrec_layer_names(l: int, prefix: str = UE):
    a = prefix + f"transformer_blocks.{l}.multi_head_attention."
    f = prefix + f"transformer_blocks.{l}.feed_forward."
    return a, f


def sasrec_prepare(p: dict, n_layers: int, dtype, prefix: str = UE, shadow: dict | None = None):
    sh = shadow or {}
    layers = []
    for l in range(n_layers):
        a, f = sasrec_layer_names(l, prefix)
        wqkv = p.get(a + "qkv_fused")   # arena view over the three adjacent projections, if the caller has one
        if wqkv is None:
            wqkv = torch.cat([p[a + "w_Q.weight"], p[a + "w_K.weight"], p[a + "w_V.weight"]], 0)
        layers.append(dict(qkv=prepare_linear(wqkv, dtype, sh.get(a + "qkv_fused"), sh.get(a + "qkv_fused" + "^T")), bqkv=None,
                           o=prepare_linear(p[a + "fc.weight"], dtype, sh.get(a + "fc.weight"), sh.get(a + "fc.weight" + "^T")), bo=None,
                           ln1_g=p[a + "layer_norm.weight"], ln1_b=p[a + "layer_norm.bias"],
                           f1=prepare_linear(p[f + "w_1.weight"], dtype, sh.get(f + "w_1.weight"), sh.get(f + "w_1.weight" + "^T")), b1=p[f + "w_1.bias"],
                           f2=prepare_linear(p[f + "w_2.weight"], dtype, sh.get(f + "w_2.weight"), sh.get(f + "w_2.weight" + "^T")), b2=p[f + "w_2.bias"],
                           ln2_g=p[f + "layer_norm.weight"], ln2_b=p[f + "layer_norm.bias"]))
    return layers


def sasrec_forward(p: dict, prep, x_in: torch.Tensor, log_mask: torch.Tensor, heads: int, need_grad: bool,
                   prefix: str = UE, drop: DropCfg = NO_DROP, res32: bool = False):
    """x_in [B, S, D] compute dtype (contiguous), log_mask float [B, S] -> [B*S, D].  ``res32`` (16-bit compute dtypes): the autocast
    data flow -- fp32 residual stream, see ``LayerCfg.res32``."""
    B, S, D = x_in.shape
    res32 = bool(res32) and ops.is16(x_in.dtype)
    cfg = LayerCfg(H=D, heads=heads, T=S, act=ACT_RELU, eps=1e-6, causal=True, mask_value=-1e9, res32=res32)
    keep = log_mask.to(torch.float32).contiguous()
    if res32:
        xh, xr, z0, mean0, rstd0 = ops.layernorm_fwd_res32(x_in.view(B * S, D), p[prefix + "layer_norm.weight"], p[prefix + "layer_norm.bias"], 1e-6,
                                                           pos=p[prefix + "position_embedding.weight"], pos_period=S,
                                                           p_out=drop.p_hidden, seed_out=drop.site(0))
        x = (xh, xr)
    else:
        x, z0, mean0, rstd0 = ops.layernorm_fwd(x_in.view(B * S, D), p[prefix + "layer_norm.weight"], p[prefix + "layer_norm.bias"],
                                                1e-6, pos=p[prefix + "position_embedding.weight"], pos_period=S,
                                                p_out=drop.p_hidden, seed_out=drop.site(0))
    saved_layers = []
    for l, w in enumerate(prep):
        x, sv = layer_forward(cfg, w, x, keep, B, need_grad, drop, 1 + 3 * l)
        saved_layers.append(sv)
    if res32:
        x = x[0]      # the user states feed the scoring GEMM: the 16-bit copy
    saved = (cfg, z0, mean0, rstd0, saved_layers, S, drop) if need_grad else None
    return x, saved


def sasrec_backward(p: dict, prep, saved, dout: torch.Tensor, grads: dict, prefix: str = UE):
    """grads: name -> fp32 buffer (accumulated).  Returns d(x_in) [B*S, D]."""
    cfg, z0, mean0, rstd0, saved_layers, S, drop = saved
    da, db = dout, None
    for l in reversed(range(len(prep))):
        a, f = sasrec_layer_names(l, prefix)
        D = cfg.H
        dqkv = grads.get(a + "qkv_fused")   # an arena may provide the fused [3D, D] block that w_Q/w_K/w_V alias
        fused = dqkv is not None
        if not fused:
            dqkv = torch.zeros((3 * D, D), device=dout.device, dtype=torch.float32)
        g = dict(qkv=dqkv, bqkv=None, o=grads[a + "fc.weight"], bo=None, ln1_g=grads[a + "layer_norm.weight"],
                 ln1_b=grads[a + "layer_norm.bias"], f1=grads[f + "w_1.weight"], b1=grads[f + "w_1.bias"],
                 f2=grads[f + "w_2.weight"], b2=grads[f + "w_2.bias"], ln2_g=grads[f + "layer_norm.weight"],
                 ln2_b=grads[f + "layer_norm.bias"])
        da, db = layer_backward(cfg, prep[l], saved_layers[l], da, db, g)
        if not fused:   # hand the three row blocks out as the parameters' gradients (views, no arithmetic)
            grads[a + "w_Q.weight"], grads[a + "w_K.weight"], grads[a + "w_V.weight"] = dqkv[:D], dqkv[D:2 * D], dqkv[2 * D:]
    dz0, _ = ops.layernorm_bwd(da, db, z0, mean0, rstd0, p[prefix + "layer_norm.weight"], grads[prefix + "layer_norm.weight"],
                               grads[prefix + "layer_norm.bias"], p_out=drop.p_hidden, seed_out=drop.site(0), sub16=False)
    ops.pos_grad_(dz0, grads[prefix + "position_embedding.weight"], S)
    return dz0      # (res32: fp32 -- the callers add it to / cast it into the item vectors' gradient)


# ---------------------------------------------------------------------------------------------------------
# BERT text encoder (T/model/encoders.py:53-70 + HF BertModel)
# ---------------------------------------------------------------------------------------------------------
TE = "bert_encoder.text_encoders.title."
# encoder layers on real tokens only (see bert_forward); MOREC_UNPAD=0 / bench.py --padded keep all T positions of every title
UNPAD_DEFAULT = os.environ.get("MOREC_UNPAD", "1") != "0"


def bert_prepare(p: dict, n_layers: int, dtype, prefix: str = TE, shadow: dict | None = None):
    sh = shadow or {}
    bm = prefix + "bert_model."
    layers = []
    for l in range(n_layers):
        L = bm + f"encoder.layer.{l}."
        wqkv, bqkv = p.get(L + "qkv_fused.weight"), p.get(L + "qkv_fused.bias")
        if wqkv is None:
            wqkv = torch.cat([p[L + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)
            bqkv = torch.cat([p[L + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)
        layers.append(dict(qkv=prepare_linear(wqkv, dtype, sh.get(L + "qkv_fused.weight"), sh.get(L + "qkv_fused.weight" + "^T")), bqkv=bqkv,
                           o=prepare_linear(p[L + "attention.output.dense.weight"], dtype, sh.get(L + "attention.output.dense.weight"), sh.get(L + "attention.output.dense.weight" + "^T")), bo=p[L + "attention.output.dense.bias"],
                           ln1_g=p[L + "attention.output.LayerNorm.weight"], ln1_b=p[L + "attention.output.LayerNorm.bias"],
                           f1=prepare_linear(p[L + "intermediate.dense.weight"], dtype, sh.get(L + "intermediate.dense.weight"), sh.get(L + "intermediate.dense.weight" + "^T")), b1=p[L + "intermediate.dense.bias"],
                           f2=prepare_linear(p[L + "output.dense.weight"], dtype, sh.get(L + "output.dense.weight"), sh.get(L + "output.dense.weight" + "^T")), b2=p[L + "output.dense.bias"],
                           ln2_g=p[L + "output.LayerNorm.weight"], ln2_b=p[L + "output.LayerNorm.bias"]))
    return dict(layers=layers, fc=prepare_linear(p[prefix + "fc.weight"], dtype, sh.get(prefix + "fc.weight"), sh.get(prefix + "fc.weight" + "^T")))


def token_packing(mask: torch.Tensor):
    """Integer bookkeeping of the unpadded layout (index arithmetic only; one host sync for the token count).
    mask int [Nc, T], a run of ones followed by zeros per row -> (cu_seqlens int32 [Nc + 1], tok_idx int32 [n_tokens]):
    sequence s owns packed rows cu[s] .. cu[s+1]-1, packed row r is padded row tok_idx[r] = s * T + t.  A row without any
    real token keeps its first position."""
    Nc, T = mask.shape
    lens = mask.sum(1).clamp_(min=1)
    cu = torch.zeros(Nc + 1, device=mask.device, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    ar = torch.arange(T, device=mask.device)
    tok_idx = (ar[None, :] < lens[:, None]).view(-1).nonzero().view(-1).to(torch.int32)
    return cu, tok_idx


def token_packing_host(mask, token_ids=None, pad_to: int = 0, pin: bool | None = None):
    """The same bookkeeping on the HOST (numpy / CPU tensor ``mask`` int [Nc, T], what the data loader's collate holds before the H2D
    copy, ``T/run.py:232-239``): returns pinned int32 CPU tensors ``(cu_seqlens [Nc + 1], tok_idx [n_tokens])`` or ``None`` when the
    rows are not a run of ones followed by zeros (the padded layout is kept then).  Uploading the two vectors with the batch spares
    the step its only two host synchronisations (the ``all()`` / ``nonzero()`` of the device-side version), i.e. ~0.25 ms of idle GPU
    at every step boundary.  With ``token_ids`` (int [Nc, T], the token half of the same rows) a third vector is returned: the rows
    of the padded layout in token-id order (stable), which the word-embedding gradient's run-length scatter walks -- otherwise a
    device-side ``argsort`` (ten small kernels) at the very end of the backward pass, with nothing left to overlap it -- and a fourth:
    padded row -> packed row (-1 for [PAD] rows), with which the backward spreads the packed gradients over the padded layout in one
    gather per tensor instead of a fill + scatter.
    ``pad_to`` > 0: ``tok_idx`` is padded with -1 up to the next multiple of ``pad_to`` (<= ``ops.SPARE_ROWS_MAX``): the packed layout then
    carries that many SPARE rows behind the last sequence -- zero rows on the way in (``morec_indexed_rows_copy`` writes zeros for a negative
    index), zero rows in every gradient, skipped by the attention kernels -- so that batches whose token counts fall into the same bucket
    have identical tensor shapes (``TrainStep.step_graphed`` replays one captured graph for all of them)."""
    import numpy as np
    m = mask.numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
    m = (m != 0)
    Nc, T = m.shape
    if T > 1 and bool((m[:, :-1] < m[:, 1:]).any()):
        return None
    lens = np.maximum(m.sum(1), 1).astype(np.int64)
    cu = np.zeros(Nc + 1, dtype=np.int32)
    cu[1:] = np.cumsum(lens)
    n = int(cu[-1])
    # packed row r of sequence s, position t -> padded row s * T + t
    seq = np.repeat(np.arange(Nc, dtype=np.int64), lens)
    pos = np.arange(n, dtype=np.int64) - np.repeat(cu[:-1].astype(np.int64), lens)
    tok = (seq * T + pos).astype(np.int32)
    if pad_to and pad_to > 0 and n % pad_to:
        assert 2 * pad_to <= ops.SPARE_ROWS_MAX, "pad_to exceeds the spare rows the attention launches zero"
        extra = pad_to - n % pad_to
        if n + extra == Nc * T:      # (a packed row count of exactly Nc T means "nothing dropped" to bert_forward: stay clear of it)
            extra += pad_to
        tok = np.concatenate((tok, np.full(extra, -1, dtype=np.int32)))
    if pin is None:      # (False from a DataLoader worker process: it must not touch the HIP runtime; the loader's pin thread page-locks)
        pin = torch.cuda.is_available()
    out = [torch.from_numpy(cu), torch.from_numpy(tok)]
    if token_ids is not None:
        ids = token_ids.numpy() if isinstance(token_ids, torch.Tensor) else np.asarray(token_ids)
        out.append(torch.from_numpy(np.argsort(ids.reshape(-1), kind="stable").astype(np.int32)))
        inv = np.full(Nc * T, -1, dtype=np.int32)      # padded row -> packed row (-1: a [PAD] row), for the way back in the backward pass
        inv[tok[:n]] = np.arange(n, dtype=np.int32)
        out.append(torch.from_numpy(inv))
    return tuple(t.pin_memory() for t in out) if pin else tuple(out)


def bert_grad_from(trainable_names, n_layers: int, prefix: str = TE) -> int:
    """Lowest point of the text tower that has a trainable parameter (the reference freezes ``bert_model`` parameters by
    index, ``T/run.py:73-75``; its default ``--freeze_paras_before 165`` = embeddings + layers 0-9, and autograd then never
    walks below layer 10): -1 = the embeddings, k >= 0 = encoder layer k, ``n_layers`` = nothing inside ``bert_model``."""
    bm = prefix + "bert_model."
    lo = n_layers
    for n in trainable_names:
        if not n.startswith(bm) or ".pooler." in n:
            continue
        if n.startswith(bm + "embeddings."):
            return -1
        if n.startswith(bm + "encoder.layer."):
            lo = min(lo, int(n[len(bm + "encoder.layer."):].split(".")[0]))
    return lo


def bert_needs_grad_buffer(name: str, grad_from: int, prefix: str = TE) -> bool:
    """Does the backward pass (which stops at ``grad_from``, see ``bert_grad_from``) write a gradient for this parameter?"""
    bm = prefix + "bert_model."
    if not name.startswith(bm):
        return True
    if ".pooler." in name:
        return False
    if name.startswith(bm + "embeddings."):
        return grad_from < 0
    if name.startswith(bm + "encoder.layer."):
        return int(name[len(bm + "encoder.layer."):].split(".")[0]) >= max(grad_from, 0)
    return True


def bert_forward(p: dict, prep, text: torch.Tensor, heads: int, dtype, need_grad: bool, eps: float = 1e-12,
                 mask_value: float = ops.FLT_MIN_MASK, prefix: str = TE, drop: DropCfg = NO_DROP, unpad: bool | None = None,
                 grad_from: int = -1, packing=None, on_use=None, res32: bool = False):
    """text int64 [Nc, 2T] = [input_ids | attention_mask] (T/model/encoders.py:63-67) -> item vectors [Nc, D].

    ``unpad``: run the encoder layers on the REAL tokens only (packed rows + ``cu_seqlens``) instead of all T positions of
    every title.  Exact for what the path consumes (``hidden[:, 0]``, encoders.py:69): [PAD] keys carry the additive
    ``finfo.min`` mask, i.e. probability exactly 0, so they never reach a real token, and every other operator is row-wise.
    Requires the mask to be a run of ones followed by zeros (what ``get_doc_input_bert``, preprocess.py:131-172, builds);
    an all-zero row (the padding item, preprocess.py:135-136) keeps its first token -- its vector reaches nothing
    (masked columns / keys, dropped rows).
    ``on_use(key)`` (optional) is called right before the first kernel that reads a parameter set -- ``"pre"`` (embeddings and whatever
    sits outside the encoder layers), ``("layer", l)``, ``"head"`` (the projection) -- so that a driver whose previous optimizer update
    is still running on another stream can make this stream wait for just that slice (``TrainStep(defer_update=True)``)."""
    if on_use is not None:
        on_use("pre")
    bm = prefix + "bert_model."
    Nc, T2 = text.shape
    T = T2 // 2
    ids32 = text[:, :T].to(torch.int32).contiguous().view(-1)
    keep = text[:, T:].to(torch.float32).contiguous()
    H = p[bm + "embeddings.word_embeddings.weight"].shape[1]
    res32 = bool(res32) and ops.is16(dtype)      # the autocast data flow: fp32 residual stream (``LayerCfg.res32``)
    cfg = LayerCfg(H=H, heads=heads, T=T, act=ACT_GELU, eps=eps, causal=False, mask_value=mask_value, res32=res32)
    type0 = p[bm + "embeddings.token_type_embeddings.weight"][0].contiguous()
    # (res32: the embedding stage runs in fp32 -- embeddings and their LayerNorm are not autocast operators -- and the first GEMM's operand
    # is its rounded copy)
    x, z_e, mean_e, rstd_e = ops.bert_embed_fwd(ids32, p[bm + "embeddings.word_embeddings.weight"],
                                                p[bm + "embeddings.position_embeddings.weight"], type0,
                                                p[bm + "embeddings.LayerNorm.weight"], p[bm + "embeddings.LayerNorm.bias"],
                                                eps, T, torch.float32 if res32 else dtype, p_out=drop.p_hidden, seed_out=drop.site(0))
    xr = None
    if res32:
        xr, x = x, ops.cast(x, dtype)
    cu, tok_idx = None, None
    n_layers = len(prep["layers"])
    order = packing[2] if packing is not None and len(packing) > 2 else None     # rows in token-id order, for the backward's word scatter
    inv = packing[3] if packing is not None and len(packing) > 3 else None       # padded row -> packed row or -1
    if packing is not None and (UNPAD_DEFAULT if unpad is None else unpad) and n_layers > 0:
        # ``packing``: (cu_seqlens, tok_idx[, order]) int32 DEVICE tensors prepared on the host with the batch (``token_packing_host``):
        # no device-side bookkeeping, no host synchronisation in the step
        cu, tok_idx = packing[0], packing[1]
        if tok_idx.numel() == Nc * T:
            cu, tok_idx = None, None
        else:
            x = ops.indexed_rows_copy(x, torch.empty((tok_idx.numel(), H), device=x.device, dtype=dtype), in_idx=tok_idx)
            if xr is not None:
                xr = ops.indexed_rows_copy(xr, torch.empty((tok_idx.numel(), H), device=x.device, dtype=torch.float32), in_idx=tok_idx)
            keep = torch.ones(tok_idx.numel(), device=x.device, dtype=torch.float32)
    elif (UNPAD_DEFAULT if unpad is None else unpad) and n_layers > 0:
        mask = text[:, T:]
        run_of_ones = bool((mask[:, :-1] >= mask[:, 1:]).all()) if T > 1 else True   # ones, then zeros (joins the sync below)
        cu, tok_idx = token_packing(mask) if run_of_ones else (None, torch.empty(Nc * T, dtype=torch.int32))
        if tok_idx.numel() == Nc * T:        # nothing to drop -- or a mask with holes, which keeps the padded layout
            cu, tok_idx = None, None          # nothing to drop
        else:
            x = ops.indexed_rows_copy(x, torch.empty((tok_idx.numel(), H), device=x.device, dtype=dtype), in_idx=tok_idx)
            if xr is not None:
                xr = ops.indexed_rows_copy(xr, torch.empty((tok_idx.numel(), H), device=x.device, dtype=torch.float32), in_idx=tok_idx)
            keep = torch.ones(tok_idx.numel(), device=x.device, dtype=torch.float32)
    if res32:
        x = (x, xr)
    saved_layers = []
    for l, w in enumerate(prep["layers"]):
        if on_use is not None:
            on_use(("layer", l))
        ng = need_grad and l >= grad_from      # layers below the first trainable one keep nothing for a backward that never reaches them
        if l == n_layers - 1:     # only hidden[:, 0] is consumed (encoders.py:69): row-wise work on the [CLS] rows only
            cls, sv = layer_forward_cls(cfg, w, x, keep, Nc, ng, drop, 1 + 3 * l, cu)
        else:
            x, sv = layer_forward(cfg, w, x, keep, Nc, ng, drop, 1 + 3 * l, cu)
        saved_layers.append(sv)
    if n_layers == 0:
        x16 = x[0] if res32 else x
        cls = ops.strided_rows_copy(x16, torch.empty((Nc, H), device=x16.device, dtype=dtype), Nc, H, T, 1)
    if on_use is not None:
        on_use("head")
    D = prep["fc"].w.shape[0]
    pre = torch.empty((Nc, D), device=cls.device, dtype=dtype) if need_grad else None
    item = ops.gemm_nt(cls, prep["fc"].w, bias=p[prefix + "fc.bias"], act=ACT_GELU, aux_out=pre)
    saved = (cfg, ids32, z_e, mean_e, rstd_e, saved_layers, cls, pre, Nc, T, H, drop, tok_idx, grad_from, order, inv) if need_grad else None
    return item, saved


def bert_backward(p: dict, prep, saved, d_item: torch.Tensor, grads: dict, prefix: str = TE, pad_id: int = 0, on_ready=None):
    """``on_ready(key)`` (optional) is called as soon as a set of gradients is final -- ``"head"`` once everything outside
    ``bert_model`` is, ``("layer", l)`` after layer ``l`` -- so a data-parallel driver can start reducing them while the
    rest of the backward pass still runs."""
    bm = prefix + "bert_model."
    cfg, ids32, z_e, mean_e, rstd_e, saved_layers, cls, pre, Nc, T, H, drop, tok_idx, grad_from, order, inv = saved
    n_layers = len(prep["layers"])
    dv = ops.act_bwd(d_item.contiguous(), pre, ACT_GELU)
    ops.colsum_(dv, grads[prefix + "fc.bias"])
    linear_wgrad_(dv, cls, grads[prefix + "fc.weight"])
    if grad_from >= n_layers and n_layers > 0:      # the whole of bert_model is frozen: the backward ends at the projection head
        if on_ready is not None:
            on_ready("head")
        return
    dcls = ops.gemm_nt(dv, prep["fc"].wt, K=dv.shape[1], N=H)
    if on_ready is not None:
        on_ready("head")
    da, db = None, None
    if n_layers == 0:
        da = torch.zeros((Nc * T, H), device=dcls.device, dtype=dcls.dtype)
        ops.strided_rows_copy(dcls, da, Nc, H, 1, T)
    for l in reversed(range(max(grad_from, 0), n_layers)):
        L = bm + f"encoder.layer.{l}."
        dqkv, dbqkv = grads.get(L + "qkv_fused.weight"), grads.get(L + "qkv_fused.bias")
        fused = dqkv is not None
        if not fused:
            dqkv = torch.zeros((3 * H, H), device=dcls.device, dtype=torch.float32)
            dbqkv = torch.zeros(3 * H, device=dcls.device, dtype=torch.float32)
        g = dict(qkv=dqkv, bqkv=dbqkv, o=grads[L + "attention.output.dense.weight"], bo=grads[L + "attention.output.dense.bias"],
                 ln1_g=grads[L + "attention.output.LayerNorm.weight"], ln1_b=grads[L + "attention.output.LayerNorm.bias"],
                 f1=grads[L + "intermediate.dense.weight"], b1=grads[L + "intermediate.dense.bias"],
                 f2=grads[L + "output.dense.weight"], b2=grads[L + "output.dense.bias"],
                 ln2_g=grads[L + "output.LayerNorm.weight"], ln2_b=grads[L + "output.LayerNorm.bias"])
        need_dx = not (grad_from >= 0 and l == grad_from)     # nothing trainable below layer grad_from (embeddings frozen too)
        if l == n_layers - 1:
            da, db = layer_backward_cls(cfg, prep["layers"][l], saved_layers[l], dcls, g, need_dx)
        else:
            da, db = layer_backward(cfg, prep["layers"][l], saved_layers[l], da, db, g, need_dx)
        if not fused:
            for i, n in enumerate(("query", "key", "value")):
                grads[L + f"attention.self.{n}.weight"] = dqkv[i * H:(i + 1) * H]
                grads[L + f"attention.self.{n}.bias"] = dbqkv[i * H:(i + 1) * H]
        if on_ready is not None:
            on_ready(("layer", l))
    if grad_from >= 0:        # frozen embeddings: no embedding-LayerNorm backward, no word / position / type scatter
        return
    if tok_idx is not None:   # back to the padded layout the embedding stage (and its dropout stream) lives in: [PAD] rows get zero
        if inv is not None:     # one gather per tensor, [PAD] rows written as zeros (no fill)
            da = ops.indexed_rows_copy(da, torch.empty((Nc * T, H), device=da.device, dtype=da.dtype), in_idx=inv)
            if db is not None:      # (res32: the residual stream's gradient is fp32, da 16-bit)
                db = ops.indexed_rows_copy(db, torch.empty((Nc * T, H), device=da.device, dtype=db.dtype), in_idx=inv)
        else:
            pa = torch.zeros((Nc * T, H), device=da.device, dtype=da.dtype)
            ops.indexed_rows_copy(da, pa, out_idx=tok_idx)
            if db is not None:
                pb = torch.zeros((Nc * T, H), device=da.device, dtype=db.dtype)
                ops.indexed_rows_copy(db, pb, out_idx=tok_idx)
                db = pb
            da = pa
    dz_e, _ = ops.layernorm_bwd(da, db, z_e, mean_e, rstd_e, p[bm + "embeddings.LayerNorm.weight"],
                                grads[bm + "embeddings.LayerNorm.weight"], grads[bm + "embeddings.LayerNorm.bias"],
                                p_out=drop.p_hidden, seed_out=drop.site(0), sub16=False)
    if order is None:      # integer bookkeeping: rows in token-id order for the run-length scatter (the collate can supply it: token_packing_host)
        order = torch.argsort(ids32, stable=ops.DETERMINISTIC).to(torch.int32)      # (stable: the fixed summation order of word_scatter_runs_kernel)
    ops.bert_embed_bwd_(ids32, dz_e, grads[bm + "embeddings.word_embeddings.weight"],
                        grads[bm + "embeddings.position_embeddings.weight"],
                        grads[bm + "embeddings.token_type_embeddings.weight"][0], pad_id, T, order)


# ---------------------------------------------------------------------------------------------------------
# in-batch debiased CE (T/model/model.py:32-33,45-67)
# ---------------------------------------------------------------------------------------------------------
@dataclass
class CeInputs:
    row_ids: torch.Tensor      # int32 [B*(S+1)]  this rank's slot ids
    col_ids: torch.Tensor      # int32 [Nc]       pool slot ids
    col_logpop: torch.Tensor   # f32   [Nc]
    col_valid: torch.Tensor    # u8    [Nc]
    row_valid: torch.Tensor    # u8    [B*S]
    B: int
    S: int
    col_offset: int


def ce_inputs_local(sample_items_id: torch.Tensor, log_mask: torch.Tensor, log_pop_table: torch.Tensor) -> CeInputs:
    """Index bookkeeping for one rank, all on device (no Python loops): column validity
    ``cat(log_mask, 1)`` (model.py:51-52), row validity (model.py:65), log-pop gather (model.py:33)."""
    B, S = log_mask.shape
    ids = sample_items_id.view(-1)
    ids32 = ids.to(torch.int32).contiguous()
    ones = torch.ones((B, 1), device=log_mask.device, dtype=log_mask.dtype)
    col_valid = (torch.cat((log_mask, ones), 1).view(-1) != 0).to(torch.uint8).contiguous()
    row_valid = (log_mask.reshape(-1) != 0).to(torch.uint8).contiguous()
    logpop = log_pop_table[ids].contiguous()
    return CeInputs(ids32, ids32, logpop, col_valid, row_valid, B, S, 0)


def ce_forward(ci: CeInputs, P: torch.Tensor, E: torch.Tensor, dE_fp32: bool = False):
    """P [B*S, D], E [Nc, D] (compute dtype).  Returns (loss_sum fp32[1] on device, saved).  ``dE_fp32``: the backward hands dE out
    in fp32 (pooled negatives: it is reduce-scattered over ranks before it is rounded to the compute dtype)."""
    dE_fp32 = bool(dE_fp32) and ops.is16(P.dtype) and E.shape[0] % 8 == 0 and P.shape[1] % 8 == 0
    desc = ops.ce_desc(ci.B, ci.S, P.shape[1], E.shape[0], ci.col_offset, P.dtype, dE_fp32)
    ws = ops.ce_workspace(desc, P.device)
    loss_sum, lse, _ = ops.inbatch_ce_fwd(desc, P, E, ci.row_ids, ci.col_ids, ci.col_logpop, ci.col_valid, ci.row_valid, ws)
    return loss_sum, (desc, ws, lse)


def ce_backward(ci: CeInputs, P, E, saved, gscale_dev, gscale: float):
    desc, ws, lse = saved
    # `ws` is the forward call's workspace, kept alive and untouched inside `saved`: the backward reuses the tables it holds
    bdesc = ops.CeDesc(desc.B, desc.S, desc.D, desc.Nc, desc.col_offset, desc.dtype, desc.dE_fp32, 1)
    return ops.inbatch_ce_bwd(bdesc, P, E, ci.row_ids, ci.col_ids, ci.col_logpop, ci.col_valid, ci.row_valid, lse,
                              gscale_dev, gscale, ws)
