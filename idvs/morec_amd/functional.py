"""``torch.autograd.Function`` shells around ``engine``: they make the HIP forward/backward a node of
PyTorch's autograd graph so that the reference driver's ``scaler.scale(loss).backward()``, ``DDP`` gradient
hooks and ``optim.AdamW`` (``T/run.py:243-247``) keep working unchanged on the drop-in ``Model``.
Gradients are computed by the hand-written backward kernels, never by autograd tracing of torch ops.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import engine, ops, swin_engine


def _zeros_like_params(names, params, skip=()):
    return {n: torch.zeros_like(p, dtype=torch.float32) for n, p in zip(names, params) if n not in skip}


class SasrecFn(torch.autograd.Function):
    """``User_Encoder.forward`` (``T/model/encoders.py:23-28``)."""

    @staticmethod
    def forward(ctx, x_in, log_mask, cfg, *params):
        names, heads, n_layers, dtype, prefix, drop = cfg[:6]
        res32 = bool(cfg[6]) if len(cfg) > 6 else False      # the autocast data flow: fp32 residual stream (engine.LayerCfg.res32)
        p = dict(zip(names, params))
        need = any(ctx.needs_input_grad)
        x = x_in.contiguous()
        if x.dtype != dtype:
            x = ops.cast(x, dtype)
        prep = engine.sasrec_prepare(p, n_layers, dtype, prefix)
        out, saved = engine.sasrec_forward(p, prep, x, log_mask, heads, need, prefix, drop, res32=res32)
        ctx.stuff = (p, prep, saved, names, prefix, x_in.dtype, tuple(x_in.shape))
        ctx.fp32_gemm = ops.FP32_GEMM
        return out.view(x_in.shape)

    @staticmethod
    def backward(ctx, dout):
        p, prep, saved, names, prefix, in_dtype, in_shape = ctx.stuff
        ctx.stuff = None
        qkv = {prefix + f"transformer_blocks.{l}.multi_head_attention.{w}.weight" for l in range(len(prep))
               for w in ("w_Q", "w_K", "w_V")}
        grads = _zeros_like_params(names, [p[n] for n in names], skip=qkv)
        d = dout.contiguous().view(-1, in_shape[-1])
        with ops.fp32_gemm_mode(ctx.fp32_gemm):
            dx = engine.sasrec_backward(p, prep, saved, d, grads, prefix)
            engine.WgradStream.join(d.device)      # autograd consumers (DDP hooks, the optimizer) read the gradients on this stream
        if dx.dtype != in_dtype:
            dx = ops.cast(dx, in_dtype)
        return (dx.view(in_shape), None, None) + tuple(grads[n] for n in names)


class BertEncoderFn(torch.autograd.Function):
    """``Text_Encoder.forward`` (``T/model/encoders.py:63-70``) over HF ``BertModel`` arithmetic."""

    @staticmethod
    def forward(ctx, text, cfg, *params):
        names, heads, n_layers, dtype, prefix, eps, mask_value, drop = cfg[:8]
        res32 = bool(cfg[8]) if len(cfg) > 8 else False
        p = dict(zip(names, params))
        need = any(ctx.needs_input_grad)
        prep = engine.bert_prepare(p, n_layers, dtype, prefix)
        # the backward stops at the lowest trainable point of the tower (T/run.py:73-75 freezes a prefix by parameter index)
        grad_from = engine.bert_grad_from([n for n, nd in zip(names, ctx.needs_input_grad[2:]) if nd], n_layers, prefix)
        item, saved = engine.bert_forward(p, prep, text, heads, dtype, need, eps, mask_value, prefix, drop, grad_from=grad_from, res32=res32)
        ctx.stuff = (p, prep, saved, names, prefix, grad_from)
        ctx.needs = ctx.needs_input_grad
        ctx.fp32_gemm = ops.FP32_GEMM
        return item

    @staticmethod
    def backward(ctx, d_item):
        p, prep, saved, names, prefix, grad_from = ctx.stuff
        ctx.stuff = None
        bm = prefix + "bert_model."
        skip = {bm + f"encoder.layer.{l}.attention.self.{n}.{k}" for l in range(len(prep["layers"]))
                for n in ("query", "key", "value") for k in ("weight", "bias")}
        skip |= {n for n in names if not engine.bert_needs_grad_buffer(n, grad_from, prefix)}    # never reached by the backward
        grads = _zeros_like_params(names, [p[n] for n in names], skip=skip)
        with ops.fp32_gemm_mode(ctx.fp32_gemm):
            engine.bert_backward(p, prep, saved, d_item.contiguous(), grads, prefix)
            engine.WgradStream.join(d_item.device)
        needs = ctx.needs[2:]
        return (None, None) + tuple(grads.get(n) if nd else None for n, nd in zip(names, needs))


class SwinEncoderFn(torch.autograd.Function):
    """``Vit_Encoder.forward`` (``V/model/encoders.py:30-31``) over HF ``SwinForImageClassification`` arithmetic."""

    @staticmethod
    def forward(ctx, pixels, cfg, *params):
        names, shape, dtype, prefix, drop, training = cfg
        p = dict(zip(names, params))
        need = any(ctx.needs_input_grad)
        prep = swin_engine.swin_prepare(p, shape, dtype, prefix)
        item, saved = swin_engine.swin_forward(p, prep, shape, pixels, dtype, need, prefix, drop, training)
        ctx.stuff = (p, prep, saved, names, prefix, shape)
        ctx.needs = ctx.needs_input_grad
        ctx.fp32_gemm = ops.FP32_GEMM
        return item

    @staticmethod
    def backward(ctx, d_item):
        p, prep, saved, names, prefix, shape = ctx.stuff
        ctx.stuff = None
        qkv = {swin_engine.swin_layer_names(prefix, s, b) + f"attention.{n}.{k}" for s, depth in enumerate(shape.depths)
               for b in range(depth) for n in ("q_proj", "k_proj", "v_proj") for k in ("weight", "bias")}
        grads = _zeros_like_params(names, [p[n] for n in names], skip=qkv)
        with ops.fp32_gemm_mode(ctx.fp32_gemm):
            swin_engine.swin_backward(p, prep, saved, d_item.contiguous(), grads, prefix)
            engine.WgradStream.join(d_item.device)
        needs = ctx.needs[2:]
        return (None, None) + tuple(grads[n] if nd else None for n, nd in zip(names, needs))


class IdEmbeddingFn(torch.autograd.Function):
    """``nn.Embedding(item_num + 1, D, padding_idx=0)`` lookup (``T/model/model.py:27,37``)."""

    @staticmethod
    def forward(ctx, ids, weight, dtype):
        idx = ids.reshape(-1).to(torch.int32).contiguous()
        out = ops.gather_rows(weight, idx, dtype)
        ctx.save_for_backward(idx)
        ctx.shape = weight.shape
        return out.view(tuple(ids.shape) + (weight.shape[1],))

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dw = torch.zeros(ctx.shape, device=dout.device, dtype=torch.float32)
        ops.scatter_add_rows_(dout.contiguous().view(-1, ctx.shape[1]), idx, dw, 0)
        return None, dw, None


def _all_gather_cat(t: torch.Tensor, world: int, comm=None) -> torch.Tensor:
    if comm is not None:       # morec_comm_all_gather: RCCL on the compute stream (comm.MorecComm)
        return comm.all_gather(t)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    if dist.get_backend() == "gloo" and t.is_cuda:   # gloo smoke tests on a GPU box: list form, staged through the host
        parts = [torch.empty_like(t, device="cpu") for _ in range(world)]
        dist.all_gather(parts, t.detach().cpu().contiguous())
        return torch.cat(parts, 0).to(t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


def pool_exchange(E: torch.Tensor, ci, n_valid: torch.Tensor, world: int, rank: int, comm=None):
    """The forward exchange of the pooled-negative step in TWO collectives instead of five: one all-gather of
    the encoded item vectors, one all-gather of a packed int32 record per rank -- slot ids | log-pop bits | validity | the
    rank's valid-row count -- whose last word also replaces the scalar all-reduce (every rank sums the same ``world`` counts in
    the same order: identical n_valid everywhere, deterministically).  Returns (E_pool, pooled CeInputs, n_valid_global)."""
    Nc = ci.col_ids.shape[0]
    blob = torch.cat((ci.col_ids.to(torch.int32), ci.col_logpop.to(torch.float32).view(torch.int32), ci.col_valid.to(torch.int32),
                      n_valid.reshape(1).to(torch.float32).view(torch.int32))).view(1, -1)
    g = _all_gather_cat(blob, world, comm)                             # [world, 3 Nc + 1]
    Epool = _all_gather_cat(E, world, comm)
    pooled = engine.CeInputs(ci.row_ids, g[:, :Nc].reshape(-1).contiguous(),
                             g[:, Nc:2 * Nc].reshape(-1).contiguous().view(torch.float32).to(ci.col_logpop.dtype),
                             g[:, 2 * Nc:3 * Nc].reshape(-1).to(torch.uint8).contiguous(), ci.row_valid, ci.B, ci.S, rank * E.shape[0])
    n_glob = g[:, 3 * Nc].contiguous().view(torch.float32).sum()
    return Epool, pooled, n_glob


def reduce_scatter_dE(dEpool: torch.Tensor, world: int, rank: int, out_dtype: torch.dtype, comm=None) -> torch.Tensor:
    """The backward exchange: SUM of every rank's gradient w.r.t. the pooled item vectors, each rank keeping the rows it owns.
    Always reduced in fp32 (an 8-way sum in bf16 would round seven times), then returned in the compute dtype."""
    t = dEpool if dEpool.dtype == torch.float32 else (ops.cast(dEpool, torch.float32) if dEpool.is_cuda else dEpool.float())
    d = comm.reduce_scatter_sum(t) if comm is not None else _reduce_scatter_sum(t, world, rank)
    if d.dtype == out_dtype:
        return d
    return ops.cast(d, out_dtype) if d.is_cuda else d.to(out_dtype)


def _reduce_scatter_sum(t: torch.Tensor, world: int, rank: int) -> torch.Tensor:
    n = t.shape[0] // world
    if dist.get_backend() == "gloo":   # gloo has no reduce_scatter: all-reduce and keep our shard (smoke / CPU tests only)
        if t.is_cuda:
            h = t.detach().float().cpu()
            dist.all_reduce(h)
            return h[rank * n:(rank + 1) * n].to(t.device).to(t.dtype).contiguous()
        dist.all_reduce(t)
        return t[rank * n:(rank + 1) * n].contiguous()
    out = torch.empty((n,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    dist.reduce_scatter_tensor(out, t.contiguous())
    return out


class InBatchCEFn(torch.autograd.Function):
    """In-batch debiased CE (``T/model/model.py:32-33,45-67``).  With ``pool=True`` under an initialised
    process group the negatives are pooled over ranks: all-gather of the encoded item
    vectors + slot ids + log-pop + validity forward, reduce-scatter(sum) of dE backward, valid-row count
    all-reduced so that N ranks x B equals the single-process loss at batch N*B.  ``loss_mult`` rescales the
    local share (``world_size`` when a gradient-AVERAGING wrapper such as DDP follows, 1 for sum-reduce)."""

    @staticmethod
    def forward(ctx, P, E, ci, pool, loss_mult, backend):
        P, E = P.contiguous(), E.contiguous()
        world = dist.get_world_size() if (pool and dist.is_available() and dist.is_initialized()) else 1
        n_valid = ci.row_valid.sum(dtype=torch.float32)
        if world > 1:
            rank = dist.get_rank()
            Epool, ci, n_valid = pool_exchange(E, ci, n_valid, world, rank)
        else:
            rank, Epool = 0, E
        loss_sum, saved = backend.ce_forward(ci, P, Epool, world > 1) if backend is engine else backend.ce_forward(ci, P, Epool)
        ctx.stuff = (ci, P, Epool, saved, world, rank, n_valid, loss_mult, backend)
        return (loss_sum[0] * loss_mult / n_valid).to(torch.float32)

    @staticmethod
    def backward(ctx, dloss):
        ci, P, Epool, saved, world, rank, n_valid, loss_mult, backend = ctx.stuff
        ctx.stuff = None
        g = (dloss.to(torch.float32) * loss_mult / n_valid).reshape(1).contiguous()
        dP, dEpool = backend.ce_backward(ci, P, Epool, saved, g, 1.0)
        dE = reduce_scatter_dE(dEpool, world, rank, dEpool.dtype) if world > 1 else dEpool
        return dP, dE, None, None, None, None
