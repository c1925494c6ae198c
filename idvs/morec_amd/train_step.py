"""``TrainStep``: the MI355X-native equivalent of the hot loop of ``T/run.py:231-247`` -- H2D batch ->
item encoder -> SASRec -> in-batch CE -> backward -> gradient reduce -> AdamW -- with no autograd graph,
no per-parameter launches and no host synchronisation inside the step.

* Parameters live in ONE flat fp32 arena per hyper-parameter group (the two groups of ``T/run.py:150-162``:
  names containing ``bert_model`` vs the rest); ``nn.Parameter.data`` of the drop-in ``Model`` are re-pointed
  at views of it, so ``state_dict()`` / checkpoints stay reference-compatible.  Q/K/V projections are laid
  out adjacently, so the fused ``[3H, H]`` QKV weight (and its gradient) is a zero-copy view.
* Gradients live in a matching flat fp32 arena (zeroed by one memset per step).  Under data parallelism the arena is
  reduced in buckets that follow the backward pass -- everything outside the tower as soon as the tower's backward starts,
  then one contiguous slice per encoder layer (Swin: per stage) as it completes -- issued asynchronously on RCCL's stream
  so the ring runs under the remaining backward kernels; one sweep at the end reduces what is left (the embeddings).
* AdamW is one kernel launch per group over the flat arena and refreshes the bf16 shadow in the same pass.
* Data parallel (one process per GPU, RCCL): negatives pooled with an all-gather of the encoded item
  vectors, dE reduce-scattered back to the owning rank, valid-row count and gradients all-reduced (SUM);
  N ranks x B is then arithmetically the single-process step at batch N*B.
"""
from __future__ import annotations

import os
import re
from collections import OrderedDict

import torch
import torch.distributed as dist

from . import engine, ops, swin_engine
from .functional import pool_exchange, reduce_scatter_dE


def _round8(n: int) -> int:
    """Arena slots start on 8-element boundaries: 32 B in the fp32 arenas, 16 B (one vector access) in the bf16 shadow."""
    return (n + 7) & ~7


class ParamArena:
    """Flat fp32 storage (+ grad, AdamW moments, optional bf16 shadow) for an ordered set of parameters."""

    def __init__(self, named_params, device, shadow):
        """``shadow``: 16-bit compute dtype of the copy the GEMMs read (``torch.bfloat16`` / ``torch.float16``) or None / False."""
        shadow = torch.bfloat16 if shadow is True else (shadow or None)
        self.offsets = OrderedDict()
        off = 0
        for name, p in named_params:
            self.offsets[name] = (off, p.numel(), tuple(p.shape))
            off += _round8(p.numel())
        self.numel = off
        self.data = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=device, dtype=torch.float32)
        self.shadow = torch.zeros(off, device=device, dtype=shadow) if shadow else None
        for name, p in named_params:
            o, n, shp = self.offsets[name]
            self.data[o:o + n].view(shp).copy_(p.data)
            p.data = self.data[o:o + n].view(shp)       # the module now reads/writes the arena
        if shadow:
            ops.cast(self.data, shadow, out=self.shadow)

    def view(self, buf, name):
        o, n, shp = self.offsets[name]
        return buf[o:o + n].view(shp)

    def span(self, buf, names, shape):
        """Zero-copy view over several ADJACENT parameters (e.g. q/k/v -> fused [3H, H])."""
        o0 = self.offsets[names[0]][0]
        end = o0
        for nme in names:
            o, n, _ = self.offsets[nme]
            assert o == end and n % 8 == 0, "parameters are not adjacent in the arena"
            end = o + n
        return buf[o0:end].view(shape)


def _check_qkv_groups(train_names, all_names, pattern, members):
    """The fused [3H, H] projection needs a layer's q / k / v parameters to be trainable TOGETHER.  A freeze boundary that
    falls inside the group (``--freeze_paras_before`` not at a layer boundary: the reference trains the unfrozen rest of the
    group, ``T/run.py:73-75``) cannot be represented by the flat arenas -- refuse it instead of silently not training them."""
    train = set(train_names)
    for n in all_names:
        mt = re.search(pattern, n)
        if not mt:
            continue
        base = n[: mt.start(1)]
        grp = [base + mbr for mbr in members]
        hit = [g_ in train for g_ in grp if g_ in all_names]
        if any(hit) and not all(hit):
            raise ValueError(f"freeze boundary inside the q/k/v group of {base!r}: freeze whole layers (parameter index at a layer "
                             "boundary, e.g. 5 + 16 k for BERT) or use the autograd path (DDP + torch.optim.AdamW) for this setting")


def _order_bert(names):
    """Arena order for the text encoder: q/k/v weights adjacent, q/k/v biases adjacent (per layer)."""
    out, seen = [], set()
    for n in names:
        if n in seen:
            continue
        if ".attention.self.query.weight" in n:
            base = n[: -len("query.weight")]
            grp = [base + "query.weight", base + "key.weight", base + "value.weight",
                   base + "query.bias", base + "key.bias", base + "value.bias"]
            out += grp
            seen.update(grp)
        elif ".attention.self." in n:
            continue
        else:
            out.append(n)
            seen.add(n)
    return out


def _order_swin(names):
    """Arena order for the Swin tower: q/k/v weights adjacent, then q/k/v biases adjacent (per block)."""
    out, seen = [], set()
    for n in names:
        if n in seen:
            continue
        if n.endswith(".attention.q_proj.weight"):
            base = n[: -len("q_proj.weight")]
            grp = [base + f"{x}_proj.{k}" for k in ("weight", "bias") for x in ("q", "k", "v")]
            out += grp
            seen.update(grp)
        elif any(n.endswith(f".attention.{x}_proj.{k}") for x in ("q", "k", "v") for k in ("weight", "bias")):
            continue
        else:
            out.append(n)
            seen.add(n)
    return out


class TrainStep:
    def __init__(self, model, *, lr: float, fine_tune_lr: float, l2_weight: float, fine_tune_l2_weight: float,
                 betas=(0.9, 0.999), eps: float = 1e-8, pool_negatives: bool = True, dedup_items: bool = False,
                 force_collectives: bool = False, comm: str | None = None, loss_scale: float | None = None,
                 dynamic_loss_scale: bool = True, growth_interval: int = 2000, defer_update: bool | None = None, graph: bool | None = None):
        """``loss_scale`` / ``dynamic_loss_scale`` / ``growth_interval``: the GradScaler of the reference's fp16 step (``T/run.py:210``:
        defaults 65536, x2 after 2000 clean steps, x0.5 and a skipped step on inf / NaN), kept in a device block (``ops.StepParams``).
        Engaged automatically for ``compute_dtype == "fp16"``; ``loss_scale=`` a number forces it on for the other dtypes too
        (``MOREC_STEP_PARAMS=1``: with scale 1) -- the device-resident step state without the scaling.
        ``defer_update`` (default: ``MOREC_DEFER_UPDATE``, off when unset -- ``run.py`` turns it ON for its ``--fused_step`` loop, whose epochs
        end in a device synchronisation before anything reads the parameters; text tower with a step block, i.e. the fp16 mode): ``step()`` leaves the AdamW launches of
        step t on the side stream, slice by slice in the order the forward pass reads the parameters, and the forward pass of step t + 1
        waits for each slice right before its first use -- the update (28 B / parameter of HBM traffic) then runs under the MFMA-bound
        encoder GEMMs of the next step instead of behind the overflow verdict at the end of its own (the other dtypes hide it under their
        own backward pass, ``_early_adamw``; a step block rules that out: the verdict covers the whole step).  CONTRACT: between two
        ``step()`` calls the parameters may still be in flight on the side stream -- read them only after ``flush()`` (which
        ``optimizer_state_dict`` / ``load_state_dict`` / ``applied_steps`` call) or a device synchronisation.
        ``graph`` (``MOREC_GRAPH=1``; one rank, no item dedup): ``step_graphed`` captures the whole step -- forward, backward, AdamW, both
        streams -- into a hipGraph per input shape and replays it (``T/run.py:231-247`` as ONE launch: the ~700 kernel launches of a step
        and the idle gaps between them go away).  Needs the step block (created with scale 1 for the non-fp16 dtypes: step count, bias
        corrections and the dropout seed word then live on the device, so no per-step host scalar is frozen into the graph)."""
        self.model = model
        # encode every DISTINCT item of the batch once (the reference re-encodes duplicates: Zipf-popular
        # items fill many of the B (S + 1) slots) and gather the vectors back to the slots; the slot gradients are
        # scatter-added before the encoder's backward.  Exact when dropout is off; with dropout the duplicates of an item
        # share one mask instead of drawing independent ones (documented deviation, hence opt-in).
        self.dedup_items = bool(dedup_items)
        self.dtype = model.compute_dtype
        self.res32 = bool(getattr(model, "res32", False))      # 16-bit GEMMs with an fp32 residual stream: the reference autocast's data flow
        self.device = next(model.parameters()).device
        self.betas, self.eps = betas, eps
        self.pool = pool_negatives
        self.step_count = 0
        named = OrderedDict((n, p) for n, p in model.named_parameters())
        train = [n for n, p in named.items() if p.requires_grad and ".pooler." not in n]
        self.vision = bool(getattr(model, "vision", False) and model.use_modal)
        _check_qkv_groups(train, list(named), r"\.attention\.self\.((?:query|key|value)\.(?:weight|bias))$",
                          [f"{x}.{k}" for x in ("query", "key", "value") for k in ("weight", "bias")])
        _check_qkv_groups(train, list(named), r"\.attention\.((?:q|k|v)_proj\.(?:weight|bias))$",
                          [f"{x}_proj.{k}" for x in ("q", "k", "v") for k in ("weight", "bias")])
        self.text_attrs = []
        if model.use_modal and not self.vision:
            # T/model/encoders.py:76-116: the token row is [ids | mask] per attribute, title | abstract | body; every attribute goes through
            # the SAME Text_Encoder and the item vector is the mean of the passes.  (name, first column, width) in that fixed order.
            enc = model.bert_encoder
            self.text_attrs = [(n, int(enc.attributes2start[n]), int(enc.attributes2length[n])) for n in ("title", "abstract", "body")
                               if n in set(enc.newsname)]
            if not self.text_attrs:
                raise ValueError(f"no text attribute among news_attributes = {list(getattr(model.args, 'news_attributes', []))}")
        if self.vision:
            # V/run.py:121-130, applied literally to the installed-HF names: 'image_net' parameters whose name contains
            # 'fc' or 'classifier' (the replaced head -- and, with transformers >= 5 naming, mlp.fc1 / mlp.fc2) train
            # with the recommender's lr / weight decay
            tower = [n for n in train if "image_net" in n and not ("fc" in n or "classifier" in n)]
            g0 = _order_swin(tower)
            g1 = [n for n in train if n not in set(tower)]
        else:
            g0 = _order_bert([n for n in train if "bert_model" in n])      # T/run.py:155: 'bert_model' in name
            g1 = [n for n in train if "bert_model" not in n]
        use_shadow = self.dtype if ops.is16(self.dtype) else None
        self.groups = []
        if g0:
            self.groups.append(dict(arena=ParamArena([(n, named[n]) for n in g0], self.device, use_shadow),
                                    lr=fine_tune_lr, wd=fine_tune_l2_weight))
        self.groups.append(dict(arena=ParamArena([(n, named[n]) for n in g1], self.device, use_shadow), lr=lr, wd=l2_weight))
        self.frozen = {n: p.data for n, p in named.items() if n not in set(g0) | set(g1)}
        self._build_views()
        self._build_transposed_shadows()
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        # ``force_collectives``: issue every collective of the data-parallel step (pooled exchange, dE reduce-scatter, bucketed
        # async gradient all-reduce, closing sweep) even on a ONE-rank group, where each of them is the identity -- so that the
        # RCCL code paths can be executed and checked on a single GPU (tests/test_train_step_rccl_gpu.py)
        if force_collectives and not (dist.is_available() and dist.is_initialized()):
            raise ValueError("force_collectives needs an initialised process group")
        self.collectives = self.world > 1 or bool(force_collectives)
        # comm = "rccl" (or MOREC_COMM=rccl): the exchange of the pooled step (two all-gathers, the dE reduce-scatter) runs through
        # the C-ABI's own RCCL communicator ON THE COMPUTE STREAM (morec_comm_*: no cross-stream events around three small
        # collectives whose results the next kernel needs at once), and the gradient buckets through a second communicator on a
        # side stream (they overlap the rest of the backward pass).  Default: torch.distributed's collectives.
        # Data parallel: the persistent GEMM grid (one 160-KiB-LDS workgroup per CU) is sized to ALL CUs; an RCCL ring kernel that
        # holds a few CUs for the length of a 28-MB bucket would push the last workgroups of every GEMM launch into a second
        # round (2x the launch).  Leave CUs out of the grid while collectives overlap the backward pass (scripts/rccl_cu_probe.py;
        # MOREC_GEMM8P_RESERVE_CUS overrides, 0 = none).
        self.overlap_reduce = os.environ.get("MOREC_OVERLAP_REDUCE", "1") != "0"
        # CUs left out of the persistent GEMM grid WHILE bucket collectives can overlap the backward pass -- set at the start of the
        # backward, cleared in reduce_gradients -- and only when a real RCCL ring exists (more than one rank, not gloo): forward, eval and
        # single-rank runs keep all 256 CUs.  Default 16 (a guess until an 8-GPU A/B run says otherwise: bench.py --sweep);
        # MOREC_GEMM8P_RESERVE_CUS (or the attribute) overrides.
        rccl_ring = (self.world > 1 and self.overlap_reduce and self.device.type == "cuda" and dist.is_initialized() and dist.get_backend() != "gloo")
        env_r = os.environ.get("MOREC_GEMM8P_RESERVE_CUS")
        self.reserve_cus = (int(env_r) if env_r is not None else 16) if rccl_ring else 0
        self._reserved = False
        comm = comm if comm is not None else os.environ.get("MOREC_COMM", "")
        self.comm = self.comm_grad = self._grad_stream = None
        if comm == "rccl" and self.collectives and self.device.type == "cuda":
            from .comm import MorecComm
            self.comm, self.comm_grad = MorecComm(), MorecComm()
            self._grad_stream = torch.cuda.Stream(device=self.device)
        self.log_pop = torch.log(model.pop_prob_list).to(self.device)
        # Device-resident step state (step count, AdamW bias corrections, loss scale, overflow flag).  fp16 activation gradients need the
        # loss scaling (their range ends at 6e-8); the other dtypes run it on request only.
        self.sp = None
        self._param_ready = {}      # deferred update: key -> event recorded behind that slice's AdamW on the side stream
        if graph is None:
            graph = os.environ.get("MOREC_GRAPH", "0") == "1"
        self.graph = bool(graph) and self.device.type == "cuda" and not self.collectives and not self.dedup_items
        if self.dtype == torch.float16 or loss_scale is not None or self.graph or os.environ.get("MOREC_STEP_PARAMS", "0") == "1":
            init = float(loss_scale) if loss_scale is not None else (65536.0 if self.dtype == torch.float16 else 1.0)
            self.sp = ops.StepParams(self.device, init_scale=init, step=0, growth_interval=growth_interval,
                                     dynamic=bool(dynamic_loss_scale) and (self.dtype == torch.float16 or loss_scale is not None))
        # A step block WITHOUT loss scaling (bf16 / fp32 under graph capture): nothing can veto the update, so the decision (step count,
        # bias corrections, next dropout seed word) is taken at the START of the step and AdamW may run bucket by bucket under the
        # backward pass as it does without a block.  With scaling the decision needs the whole step's gradients and comes last.
        self._decide_first = self.sp is not None and not self.sp.dynamic and init == 1.0
        self._decided = False
        # one captured graph (+ its static inputs and its share of the graph memory pool) per input shape, at most ``graph_max`` of them
        # (MOREC_GRAPH_MAX, default 8): shapes beyond that run as plain eager steps instead of growing the pool without bound
        self._graphs, self._graph_pool = OrderedDict(), None
        self.graph_max = max(1, int(os.environ.get("MOREC_GRAPH_MAX", "8")))
        self.buckets = self._bucket_plan()
        if defer_update is None:
            defer_update = os.environ.get("MOREC_DEFER_UPDATE", "0") == "1"
        self.defer_update = (bool(defer_update) and self.sp is not None and not self._decide_first and model.use_modal and not self.vision
                             and self.device.type == "cuda")
        self._pending, self._reduced, self._stepped = [], [], []
        self._fused_update = False
        self._in_step = False
        self.trace = None           # list -> _reduce_slice / reduce_gradients record stream-time events of the gradient collectives
        self._inflight = []         # end-of-step events of the steps the host has issued and not yet waited for (at most two)

    def _bucket_plan(self):
        """Contiguous slices of the tower arena whose gradients become final together: ``("layer", l)`` for the text
        encoder, ``("stage", s)`` for Swin (keys of the engines' ``on_ready`` callback).  Parameters outside any slice
        (embeddings, final norm) are reduced by the closing sweep of ``reduce_gradients``."""
        if len(self.groups) < 2:
            return {}
        a0 = self.groups[0]["arena"]
        pat = re.compile(r"\.encoder\.layers\.(\d+)\." if self.vision else r"\.encoder\.layer\.(\d+)\.")
        tag = "stage" if self.vision else "layer"
        spans = {}
        for n, (o, cnt, _) in a0.offsets.items():
            mt = pat.search(n)
            if mt:
                lo, hi, tot = spans.get(int(mt.group(1)), (o, o, 0))
                spans[int(mt.group(1))] = (min(lo, o), max(hi, o + _round8(cnt)), tot + _round8(cnt))
        # a slice is only usable if nothing else sits inside it
        return {(tag, k): (lo, hi) for k, (lo, hi, tot) in spans.items() if hi - lo == tot}

    # -----------------------------------------------------------------------------------------------
    def _build_views(self):
        m = self.model
        self.p, self.g, self.sh = {}, {}, {}
        for grp in self.groups:
            a = grp["arena"]
            for n in a.offsets:
                self.p[n] = a.view(a.data, n)
                self.g[n] = a.view(a.grad, n)
                if a.shadow is not None:
                    self.sh[n] = a.view(a.shadow, n)
        self.p.update(self.frozen)
        D = m.args.embedding_dim
        ue = engine.UE
        a1 = self.groups[-1]["arena"]
        for l in range(m.args.transformer_block):
            a, _ = engine.sasrec_layer_names(l, ue)
            names = [a + "w_Q.weight", a + "w_K.weight", a + "w_V.weight"]
            self.p[a + "qkv_fused"] = a1.span(a1.data, names, (3 * D, D))
            self.g[a + "qkv_fused"] = a1.span(a1.grad, names, (3 * D, D))
            if a1.shadow is not None:
                self.sh[a + "qkv_fused"] = a1.span(a1.shadow, names, (3 * D, D))
        if self.vision:
            a0 = self.groups[0]["arena"]
            self.swin_shape = m.cv_encoder.image_net.shape
            C = self.swin_shape.embed_dim
            for s_, depth in enumerate(self.swin_shape.depths):
                for b in range(depth):
                    A = swin_engine.swin_layer_names(swin_engine.IN, s_, b) + "attention."
                    wn = [A + f"{x}_proj.weight" for x in ("q", "k", "v")]
                    bn = [A + f"{x}_proj.bias" for x in ("q", "k", "v")]
                    if wn[0] not in a0.offsets:   # frozen block (V/run.py:58-60): per-step concatenation instead
                        continue
                    self.p[A + "qkv_fused.weight"] = a0.span(a0.data, wn, (3 * C, C))
                    self.p[A + "qkv_fused.bias"] = a0.span(a0.data, bn, (3 * C,))
                    self.g[A + "qkv_fused.weight"] = a0.span(a0.grad, wn, (3 * C, C))
                    self.g[A + "qkv_fused.bias"] = a0.span(a0.grad, bn, (3 * C,))
                    if a0.shadow is not None:
                        self.sh[A + "qkv_fused.weight"] = a0.span(a0.shadow, wn, (3 * C, C))
                C *= 2
        elif m.use_modal:
            a0 = self.groups[0]["arena"]
            bert = m.bert_encoder.text_encoders["title"].bert_model
            H, L = bert.config.hidden_size, bert.config.num_hidden_layers
            self.bert_heads = bert.config.num_attention_heads
            self.bert_layers, self.bert_eps = L, bert.config.layer_norm_eps
            self.bert_mask_value = m.bert_encoder.text_encoders["title"].mask_value
            # lowest trainable point of the tower: the backward (and what the forward keeps for it) stops there
            self.bert_grad_from = engine.bert_grad_from(list(a0.offsets) if len(self.groups) > 1 else [], L)
            for l in range(L):
                Lp = engine.TE + f"bert_model.encoder.layer.{l}."
                wn = [Lp + f"attention.self.{n}.weight" for n in ("query", "key", "value")]
                bn = [Lp + f"attention.self.{n}.bias" for n in ("query", "key", "value")]
                if wn[0] not in a0.offsets:   # frozen layer: fall back to per-step concatenation
                    continue
                self.p[Lp + "qkv_fused.weight"] = a0.span(a0.data, wn, (3 * H, H))
                self.p[Lp + "qkv_fused.bias"] = a0.span(a0.data, bn, (3 * H,))
                self.g[Lp + "qkv_fused.weight"] = a0.span(a0.grad, wn, (3 * H, H))
                self.g[Lp + "qkv_fused.bias"] = a0.span(a0.grad, bn, (3 * H,))
                if a0.shadow is not None:
                    self.sh[Lp + "qkv_fused.weight"] = a0.span(a0.shadow, wn, (3 * H, H))

    def _build_transposed_shadows(self):
        """Persistent W^T copies (compute dtype) of every Linear weight the engines' ``*_prepare`` ask a shadow for, refreshed by
        ONE ``morec_transpose_batch`` launch per step instead of one ``morec_transpose`` launch and allocation per weight (60 per
        step at BERT-base).  ``self.sh[name + "^T"]`` is what ``engine.prepare_linear`` picks up.  bf16 mode only (no shadows in
        the exact-fp32 mode: it keeps the per-weight path)."""
        self._wt_batch = None
        if not self.sh:
            return
        m, asked = self.model, []

        class _Recorder(dict):
            def get(self, key, default=None):
                if not key.endswith("^T"):
                    asked.append(key)
                return dict.get(self, key, default)

        rec = _Recorder(self.sh)
        if self.vision:
            swin_engine.swin_prepare(self.p, self.swin_shape, self.dtype, swin_engine.IN, rec)
        elif m.use_modal:
            engine.bert_prepare(self.p, self.bert_layers, self.dtype, engine.TE, rec)
        engine.sasrec_prepare(self.p, m.args.transformer_block, self.dtype, engine.UE, rec)
        todo, total = [], 0
        for k in dict.fromkeys(asked):
            t = self.sh.get(k)
            if t is None or t.dim() != 2 or t.dtype != self.dtype or not t.is_contiguous():
                continue
            out_f, in_f = t.shape
            ld = (out_f + 7) // 8 * 8
            todo.append((k, t, in_f, ld, total))
            total += (in_f * ld + 7) // 8 * 8
        if not todo:
            return
        store = torch.zeros(total, device=self.device, dtype=self.dtype)      # pad columns stay zero
        pairs = []
        for k, t, in_f, ld, off in todo:
            dst = store[off:off + in_f * ld].view(in_f, ld)
            if ops.TransposeBatch.eligible(t, dst):
                self.sh[k + "^T"] = dst
                pairs.append((t, dst))
        if pairs:
            self._wt_store = store
            self._wt_batch = ops.TransposeBatch(pairs)

    # -----------------------------------------------------------------------------------------------
    def forward_backward(self, sample_items_id, sample_items, log_mask, token_packing=None):
        with ops.fp32_gemm_mode(getattr(self.model, "fp32_gemm", "exact")):      # how fp32 GEMMs run for this model ("exact" | "bf16x3")
            return self._forward_backward(sample_items_id, sample_items, log_mask, token_packing)

    def _forward_backward(self, sample_items_id, sample_items, log_mask, token_packing=None):
        """One forward + backward into the gradient arenas.  Returns the loss (device scalar, no sync).  Under data
        parallelism the bucketed gradient reduction is STARTED here (async collectives issued from the backward pass);
        ``reduce_gradients`` must follow to reduce the rest and join them before the arenas are read."""
        m, p, g = self.model, self.p, self.g
        D, S = m.args.embedding_dim, m.max_seq_len
        self._begin_step()
        # Nothing in the FORWARD pass reads the gradient arenas or the W^T copies (dX = dY W): zero / refresh them on the side stream,
        # under the forward GEMMs (HBM-bound fills next to MFMA-bound kernels), and let the main stream wait for them right before
        # the backward pass starts.
        # So does the scoring stage's index bookkeeping (validity masks, log-popularity gather, 1 / n_valid: a dozen small kernels that
        # depend on the batch's ids and mask only) -- between the two towers they would sit on the critical path at ~6 us apiece.
        ids = sample_items_id.view(-1)
        side = engine.WgradStream.get(self.device)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(self.device))     # the previous step's AdamW wrote the shadows the transposes read
            with torch.cuda.stream(side):
                ci = engine.ce_inputs_local(ids, log_mask, self.log_pop)
                n_valid = ci.row_valid.sum(dtype=torch.float32)
                gscale = self._gscale(n_valid)
                self._prepare_step_buffers()
            engine.WgradStream._dirty.add(self.device)
        else:
            ci = engine.ce_inputs_local(ids, log_mask, self.log_pop)
            n_valid = ci.row_valid.sum(dtype=torch.float32)
            gscale = self._gscale(n_valid)
            self._prepare_step_buffers()
        self._pending, self._reduced, self._stepped = [], [], []
        # gradient dict handed to the engine: arena views; frozen tensors get scratch buffers
        grads = dict(g)
        for n, t in self.frozen.items():      # frozen tensors the backward still passes through get scratch buffers
            if ".pooler." not in n and (self.vision or not m.use_modal or engine.bert_needs_grad_buffer(n, self.bert_grad_from)):
                grads[n] = torch.zeros_like(t)
        d_item, d_user = m.dropout_cfgs()
        dedup = self.dedup_items and m.use_modal
        if dedup:   # integer bookkeeping only (sort / unique / first-occurrence index); one host sync for the count
            uniq, inv = torch.unique(ids, return_inverse=True)
            first = torch.empty(uniq.shape[0], device=ids.device, dtype=torch.long)
            first.scatter_(0, inv, torch.arange(ids.shape[0], device=ids.device))   # any occurrence: same item, same content
            slot_items, sample_items = sample_items, sample_items[first].contiguous()
            inv32 = inv.to(torch.int32).contiguous()
        if self.vision:
            prep_b = swin_engine.swin_prepare(p, self.swin_shape, self.dtype, swin_engine.IN, self.sh)
            E, saved_b = swin_engine.swin_forward(p, prep_b, self.swin_shape, sample_items, self.dtype, True, swin_engine.IN,
                                                  d_item, m.training)
        elif m.use_modal:
            width = sum(w for _, _, w in self.text_attrs)
            if sample_items.shape[1] != width:
                raise ValueError(f"token rows of width {sample_items.shape[1]}: expected [input_ids | attention_mask] per attribute "
                                 f"({', '.join(f'{n}: {w}' for n, _, w in self.text_attrs)}) = {width}")
            prep_b = engine.bert_prepare(p, self.bert_layers, self.dtype, engine.TE, self.sh)
            n_attr = len(self.text_attrs)
            # ``token_packing``: one (cu, tok[, order, inv]) tuple for the single attribute, a sequence of such tuples (or None) per attribute otherwise
            packs = [None] * n_attr if (dedup or token_packing is None) else ([token_packing] if n_attr == 1 else list(token_packing))
            if len(packs) != n_attr:
                raise ValueError(f"token_packing: {len(packs)} entries for {n_attr} text attributes")
            passes = []
            for ai, (_, a0, aw) in enumerate(self.text_attrs):      # every attribute through the SAME encoder (T/model/encoders.py:107-112)
                sub = sample_items if n_attr == 1 else sample_items[:, a0:a0 + aw].contiguous()
                passes.append(engine.bert_forward(p, prep_b, sub, self.bert_heads, self.dtype, True, self.bert_eps,
                                                  self.bert_mask_value, engine.TE, d_item.stream(ai), grad_from=self.bert_grad_from,
                                                  packing=packs[ai], on_use=self._await_params if (self._param_ready and ai == 0) else None,
                                                  res32=self.res32))
            E = passes[0][0] if n_attr == 1 else ops.scaled_sum([e for e, _ in passes], 1.0 / n_attr)      # encoders.py:113-116: the mean
            self._await_params(None)      # whatever the tower did not ask for (the recommender group's slice) before SASRec reads it
        else:
            idx32 = sample_items.view(-1).to(torch.int32).contiguous()
            E = ops.gather_rows(p["id_embedding.weight"], idx32, self.dtype)
        if dedup:   # distinct-item vectors -> slot vectors (row gather through the fp32 staging the gather kernel reads)
            E_u = E
            E = ops.gather_rows(ops.cast(E_u, torch.float32) if E_u.dtype != torch.float32 else E_u, inv32, self.dtype)
        B = log_mask.shape[0]
        x_in = E.view(B, S + 1, D)[:, :-1, :].contiguous()
        prep_s = engine.sasrec_prepare(p, m.args.transformer_block, self.dtype, engine.UE, self.sh)
        P, saved_s = engine.sasrec_forward(p, prep_s, x_in, log_mask, m.args.num_attention_heads, True, engine.UE, d_user, res32=self.res32)
        if side is not None:       # scoring bookkeeping done, gradient arenas zeroed, W^T copies in place
            torch.cuda.current_stream(self.device).wait_stream(side)
        Epool = E
        if self.collectives and self.pool:      # two collectives: item vectors + one packed (ids | log-pop | validity | n_valid) record
            Epool, ci, n_valid = pool_exchange(E, ci, n_valid, self.world, self.rank, self.comm)
            gscale = self._gscale(n_valid)
        loss_sum, saved_c = engine.ce_forward(ci, P, Epool, dE_fp32=(self.collectives and self.pool))
        self._reserve(True)      # the backward pass starts: bucket collectives may run beside its GEMMs from here on
        dP, dEpool = engine.ce_backward(ci, P, Epool, saved_c, gscale, 1.0)
        dE = reduce_scatter_dE(dEpool, self.world, self.rank, self.dtype, self.comm) if (self.collectives and self.pool) else dEpool
        dx = engine.sasrec_backward(p, prep_s, saved_s, dP, grads, engine.UE)
        dE.view(B, S + 1, D)[:, :-1, :].add_(dx.view(B, S, D))      # the two sources of dE (T/model/model.py:39-41,49)
        if dedup:   # slot gradients -> distinct-item gradients (fp32 accumulation), back to the compute dtype for the encoder
            dE_u = torch.zeros((E_u.shape[0], D), device=dE.device, dtype=torch.float32)
            ops.scatter_add_rows_(dE.contiguous(), inv32, dE_u, -1)
            dE = dE_u if self.dtype == torch.float32 else ops.cast(dE_u, self.dtype)
        if self.vision:
            swin_engine.swin_backward(p, prep_b, saved_b, dE, grads, swin_engine.IN, on_ready=self._on_ready)
        elif m.use_modal:
            # mean over attributes: each pass receives dE / n; the passes share every parameter, their gradients ACCUMULATE in the arenas, so
            # a bucket is final -- reduced over ranks / stepped -- only behind the LAST pass
            d_pass = dE if len(passes) == 1 else ops.scaled_sum([dE.contiguous()], 1.0 / len(passes))
            for ai, (_, saved_a) in enumerate(passes):
                engine.bert_backward(p, prep_b, saved_a, d_pass, grads, engine.TE, on_ready=self._on_ready if ai == len(passes) - 1 else None)
        else:
            ops.scatter_add_rows_(dE, idx32, grads["id_embedding.weight"], 0)
        engine.WgradStream.join(self.device)       # the weight gradients of the side stream are final from here on
        ops.x3_cache_clear()
        return loss_sum[0] / n_valid

    def _begin_step(self):
        """Static step block: step count + 1, bias corrections and the dropout seed word of THIS step, once per step (a no-op otherwise)."""
        if self._decide_first and not self._decided:
            self.sp.decide_(self.betas[0], self.betas[1])
            self._decided = True

    def _gscale(self, n_valid):
        """Device scalar the loss gradient starts from: 1 / n_valid, times the loss scale of the step block when there is one."""
        if self.sp is None:
            return (1.0 / n_valid).reshape(1)
        return (self.sp.loss_scale_dev / n_valid).reshape(1)

    def _reserve(self, on: bool):
        """Launch-time knob of the persistent GEMM grid (``morec_tuning_set("gemm8p_reserve_cus")``): ``reserve_cus`` CUs stay free for the
        RCCL ring kernel between the start of the backward pass and the join of the gradient collectives, none outside that window."""
        if self.reserve_cus <= 0 or on == self._reserved:
            return
        from . import _lib
        _lib.lib().morec_tuning_set(b"gemm8p_reserve_cus", self.reserve_cus if on else 0)
        self._reserved = on

    def _prepare_step_buffers(self):
        if self._wt_batch is not None:
            self._wt_batch.run()          # W^T of every Linear weight from the shadows AdamW (or sync_shadow) last wrote
        for grp in self.groups:
            grp["arena"].grad.zero_()

    @staticmethod
    def _all_reduce(t):
        if dist.get_backend() == "gloo" and t.is_cuda:   # functional smoke test of the N > 1 path on one GPU
            h = t.detach().cpu()
            dist.all_reduce(h)
            t.copy_(h)
        else:
            dist.all_reduce(t)

    def _reduce_slice(self, gi, lo, hi):
        t = self.groups[gi]["arena"].grad[lo:hi]
        if self.trace is not None:      # diagnostics (bench.py, N > 1): when, in stream time, each bucket's collective is issued
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self.trace.append(("issue", gi, lo, hi, ev))
        if self.comm_grad is not None:     # own RCCL communicator on a side stream, behind the kernels already queued on this one
            self._grad_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._grad_stream):
                self.comm_grad.all_reduce_sum_(t)
                if self.trace is not None:      # completion of this bucket's collective in stream time (own communicator only)
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(self._grad_stream)
                    self.trace.append(("done", gi, lo, hi, ev))
            self._pending.append(None)
        elif dist.get_backend() == "gloo":
            self._all_reduce(t)
        else:   # RCCL: enqueued behind the kernels already on this stream, runs on the communicator's own stream
            self._pending.append(dist.all_reduce(t, async_op=True))
        self._reduced.append((gi, lo, hi))

    def _on_ready(self, key):
        """Backward-pass callback of the engines: start reducing the gradients that have just become final."""
        if not self.collectives:
            self._early_adamw(key)
            return
        if not self.overlap_reduce:
            return
        side = engine.WgradStream.get(self.device) if self.device in engine.WgradStream._dirty else None
        if side is not None:      # this bucket's weight gradients are being written on the weight-gradient stream: order the collective behind THAT stream
            side.wait_stream(torch.cuda.current_stream(self.device))    # ... and behind the main stream's share (LayerNorm / bias gradients)
            with torch.cuda.stream(side):
                self._issue_ready(key)
            return
        self._issue_ready(key)

    def _early_adamw(self, key):
        """Single process, inside ``step()``: the AdamW update of a bucket whose gradients are final (the recommender group once the
        tower's backward is under way, an encoder layer / Swin stage when its backward is done) runs on the side stream behind that
        bucket's weight-gradient GEMMs, UNDER the rest of the backward pass (28 B / parameter of HBM traffic next to MFMA-bound
        GEMMs) instead of after it.  Safe: nothing later in this step reads these weights again -- the backward walks the layers
        downwards -- and the bf16 shadow / W^T copies are only read by the next step."""
        if not self._fused_update or (self.sp is not None and not self._decide_first):      # with loss scaling the update waits for the overflow verdict of the WHOLE step
            return
        side = engine.WgradStream.get(self.device)
        if side is None:
            return
        if key == "head":
            gi = len(self.groups) - 1
            lo, hi = 0, self.groups[gi]["arena"].numel
        elif key in self.buckets:
            gi, (lo, hi) = 0, self.buckets[key]
        else:
            return
        side.wait_stream(torch.cuda.current_stream(self.device))      # LayerNorm / bias gradients of the bucket are written on the main stream
        with torch.cuda.stream(side):
            self._adamw_slice(gi, lo, hi, self.step_count + 1)
        engine.WgradStream._dirty.add(self.device)
        self._stepped.append((gi, lo, hi))

    def _adamw_slice(self, gi, lo, hi, step):
        grp = self.groups[gi]
        a = grp["arena"]
        sh = None if a.shadow is None else a.shadow[lo:hi]
        if self.sp is not None:      # step count / bias corrections from the device block (decided at the start of this step)
            ops.adamw_sp_(a.data[lo:hi], a.grad[lo:hi], a.exp_avg[lo:hi], a.exp_avg_sq[lo:hi], sh, grp["lr"], self.betas[0], self.betas[1], self.eps,
                          grp["wd"], self.sp)
            return
        ops.adamw_(a.data[lo:hi], a.grad[lo:hi], a.exp_avg[lo:hi], a.exp_avg_sq[lo:hi], sh, grp["lr"], self.betas[0], self.betas[1], self.eps,
                   grp["wd"], step)

    def _issue_ready(self, key):
        if key == "head":      # the recommender group (SASRec, fc, id table) is complete once the tower's backward is under way
            gi = len(self.groups) - 1
            self._reduce_slice(gi, 0, self.groups[gi]["arena"].numel)
        elif key in self.buckets:
            self._reduce_slice(0, *self.buckets[key])

    def reduce_gradients(self):
        """SUM over ranks (the 1/n_valid_global factor is already inside the loss gradient): closes the bucketed
        reduction started during the backward pass -- reduces every slice not yet issued, then joins the async work."""
        if self.collectives:
            for gi, grp in enumerate(self.groups):
                done = sorted((lo, hi) for g_, lo, hi in self._reduced if g_ == gi)
                pos = 0
                for lo, hi in done + [(grp["arena"].numel, grp["arena"].numel)]:
                    assert lo >= pos, "overlapping gradient buckets"
                    if lo > pos:
                        self._reduce_slice(gi, pos, lo)
                    pos = hi
            if self.trace is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.device))
                self.trace.append(("join_begin", ev))
            for w in self._pending:
                if w is not None:
                    w.wait()
            if self._grad_stream is not None:
                torch.cuda.current_stream().wait_stream(self._grad_stream)
            self._pending = []
            if self.trace is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.device))
                self.trace.append(("join_end", ev))
            if not self.pool:   # rank-local negatives: the reference's DDP MEAN over ranks (T/run.py:148)
                for grp in self.groups:
                    grp["arena"].grad.mul_(1.0 / self.world)
        self._reserve(False)

    # -----------------------------------------------------------------------------------------------
    def sync_shadow(self):
        """Re-derive the bf16 shadow of every arena from its fp32 master.  The shadow is otherwise written only at
        construction and by the AdamW kernel, and the Linear weights of the next forward / backward are read from it: call
        this after ANY in-place write to the parameters that did not go through ``optimizer_step`` -- ``load_state_dict``
        (checkpoint resume), manual re-initialisation, a rank-0 broadcast."""
        for grp in self.groups:
            a = grp["arena"]
            if a.shadow is not None:
                ops.cast(a.data, a.shadow.dtype, out=a.shadow)

    def load_state_dict(self, state_dict, strict: bool = True):
        """``model.load_state_dict`` + ``sync_shadow`` (the parameters are views of the arenas, so the copy lands there)."""
        self.flush()
        out = self.model.load_state_dict(state_dict, strict=strict)
        self.sync_shadow()
        return out

    def _reference_param_order(self):
        """Trainable parameters in the order of the reference's two AdamW groups (``T/run.py:150-162``: names containing
        ``bert_model`` first, the rest second, each in ``named_parameters()`` order; V/run.py:121-135 for the vision tower)."""
        named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad and ".pooler." not in n]
        if self.vision:
            tower = lambda n: "image_net" in n and not ("fc" in n or "classifier" in n)   # noqa: E731
        else:
            tower = lambda n: "bert_model" in n                                            # noqa: E731
        g0 = [n for n, _ in named if tower(n)]
        g1 = [n for n, _ in named if not tower(n)]
        return [g for g in (g0, g1) if g]

    def _arena_of(self, name):
        for grp in self.groups:
            if name in grp["arena"].offsets:
                return grp
        raise KeyError(name)

    def optimizer_state_dict(self):
        """The fused optimizer's state in the form of ``torch.optim.AdamW.state_dict()`` for the reference's parameter
        groups, so that ``save_model`` / ``load_model`` (``T/data_utils/utils.py:107-114``, ``T/run.py:193-195``) and a plain
        ``optim.AdamW`` can exchange checkpoints with a ``--fused_step`` run."""
        state, groups, idx = {}, [], 0
        n_applied = self.applied_steps()
        for names in self._reference_param_order():
            grp = self._arena_of(names[0])
            ids = []
            for n in names:
                a = self._arena_of(n)["arena"]
                state[idx] = {"step": torch.tensor(float(n_applied)), "exp_avg": a.view(a.exp_avg, n).detach().clone(),
                              "exp_avg_sq": a.view(a.exp_avg_sq, n).detach().clone()}
                ids.append(idx)
                idx += 1
            groups.append({"lr": grp["lr"], "betas": tuple(self.betas), "eps": self.eps, "weight_decay": grp["wd"], "amsgrad": False,
                           "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                           "params": ids})
        return {"state": state, "param_groups": groups}

    def load_optimizer_state_dict(self, sd):
        order = [n for names in self._reference_param_order() for n in names]
        ids = [i for g_ in sd["param_groups"] for i in g_["params"]]
        if len(ids) != len(order):
            raise ValueError(f"optimizer state holds {len(ids)} parameters, this model trains {len(order)}")
        steps = set()
        for i, n in zip(ids, order):
            st = sd["state"].get(i)
            if st is None:          # parameter never stepped
                continue
            a = self._arena_of(n)["arena"]
            a.view(a.exp_avg, n).copy_(st["exp_avg"].to(a.exp_avg.device))
            a.view(a.exp_avg_sq, n).copy_(st["exp_avg_sq"].to(a.exp_avg.device))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused AdamW keeps one step count")
        self.step_count = steps.pop() if steps else 0
        if self.sp is not None:      # the device block carries the step count the bias corrections are formed from
            h = self.sp.host()
            self.sp.i32[0:1].fill_(self.step_count)
            del h

    def applied_steps(self) -> int:
        """Optimizer steps that really updated the parameters (synchronises when a step block is in use: steps skipped for a non-finite
        gradient do not count, exactly as torch's AdamW ``step`` under GradScaler)."""
        self.flush()
        return int(self.sp.host().step) if self.sp is not None else self.step_count

    def scaler_state_dict(self):
        """``GradScaler.state_dict()`` of the device block (``T/run.py`` never saves it; offered for exact resumption)."""
        if self.sp is None:
            return {}
        h = self.sp.host()
        return {"scale": float(h.loss_scale), "growth_factor": self.sp.growth_factor, "backoff_factor": self.sp.backoff_factor,
                "growth_interval": self.sp.growth_interval, "_growth_tracker": int(h.growth_tracker), "dynamic": bool(self.sp.dynamic)}

    def load_scaler_state_dict(self, sd):
        """Counterpart of ``scaler_state_dict`` (``GradScaler.load_state_dict``): the loss scale and the count of clean steps since its
        last change go back into the device block, so a resumed fp16 run does not restart at 65536 and skip its first steps.  An empty
        dict (a checkpoint of the reference, which never saves its scaler) leaves the block as it is."""
        if self.sp is None or not sd:
            return
        if not self.sp.dynamic or not sd.get("dynamic", True):
            # a scale saved by a static block (bf16 / fp32 step: 1.0) says nothing about fp16 overflow, and a static block has no use for a
            # dynamic run's scale: keep this run's own
            import warnings
            warnings.warn("load_scaler_state_dict: the checkpoint's loss scale comes from a different loss-scaling mode "
                          f"(saved dynamic={sd.get('dynamic', True)}, this step dynamic={self.sp.dynamic}); ignored")
            return
        self.flush()
        scale = float(sd["scale"])
        if not scale > 0.0:
            raise ValueError(f"loss scale {scale}")
        self.sp.f32[4:5].fill_(scale)
        self.sp.f32[5:6].fill_(1.0 / scale)
        self.sp.i32[2:3].fill_(int(sd.get("_growth_tracker", 0)))
        self.sp.growth_factor = float(sd.get("growth_factor", self.sp.growth_factor))
        self.sp.backoff_factor = float(sd.get("backoff_factor", self.sp.backoff_factor))
        self.sp.growth_interval = int(sd.get("growth_interval", self.sp.growth_interval))

    def global_loss(self, loss):
        """With pooled negatives ``step`` returns THIS rank's share ``loss_sum_local / n_valid_global`` (the shares add up to
        the loss of the single-process step at batch N*B); this is the SUM over ranks, for logging."""
        if self.collectives and self.pool:
            t = loss.detach().float().clone().reshape(1)
            if self.comm is not None:
                self.comm.all_reduce_sum_(t)
            else:
                self._all_reduce(t)
            return t[0]
        return loss

    def _await_params(self, key):
        """Deferred update: make the CURRENT stream wait for the AdamW of the slice the forward pass is about to read (``key`` None: all
        that is left)."""
        if not self._param_ready:
            return
        cur = torch.cuda.current_stream(self.device)
        for k in ([key] if key is not None else list(self._param_ready)):
            ev = self._param_ready.pop(k, None)
            if ev is not None:
                cur.wait_event(ev)

    def flush(self):
        """Deferred update: the current stream waits for every AdamW launch still in flight on the side stream.  Call before reading
        parameters / optimizer state between steps (a device synchronisation does the same)."""
        self._await_params(None)

    def _update_plan(self):
        """Slices of the arenas in the order the forward pass first reads them: everything of the tower group outside the per-layer
        buckets (the embeddings) -> "pre", the encoder layers, then the recommender group (projection head + SASRec) -> "head"."""
        plan = []
        if len(self.groups) > 1:
            n0 = self.groups[0]["arena"].numel
            layers = sorted((lo, hi, k) for k, (lo, hi) in self.buckets.items())
            pos = 0
            for lo, hi, k in layers:
                if lo > pos:
                    plan.append(("pre", 0, pos, lo))
                pos = hi
            if pos < n0:
                plan.append(("pre", 0, pos, n0))
            plan = [x for x in plan if x[0] == "pre"] + [(k, 0, lo, hi) for lo, hi, k in layers]
        gi = len(self.groups) - 1
        plan.append(("head", gi, 0, self.groups[gi]["arena"].numel))
        return plan

    def optimizer_step(self):
        """AdamW over everything ``_early_adamw`` has not already stepped during this step's backward pass (all of it outside ``step()``).
        With a step block (fp16 mode): GradScaler.step + update on the device -- overflow check of every gradient arena, the decision,
        then the update (or nothing); ``step_count`` then counts the calls, ``applied_steps()`` the updates that happened."""
        self.step_count += 1
        if self._decide_first:
            self._begin_step()      # (a direct call without forward_backward: the block still has to move on)
            self._decided = False
        elif self.sp is not None:
            self.flush()      # (a deferred update of the previous step that nothing has waited for yet)
            for grp in self.groups:
                self.sp.check_finite_(grp["arena"].grad)
            self.sp.decide_(self.betas[0], self.betas[1])
            side = engine.WgradStream.get(self.device) if (self.defer_update and self._in_step) else None
            if side is None:
                for grp in self.groups:
                    a = grp["arena"]
                    ops.adamw_sp_(a.data, a.grad, a.exp_avg, a.exp_avg_sq, a.shadow, grp["lr"], self.betas[0], self.betas[1], self.eps, grp["wd"], self.sp)
            else:      # deferred: slice by slice in forward order on the side stream, an event behind each (see __init__)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    for key, gi, lo, hi in self._update_plan():
                        grp = self.groups[gi]
                        a = grp["arena"]
                        ops.adamw_sp_(a.data[lo:hi], a.grad[lo:hi], a.exp_avg[lo:hi], a.exp_avg_sq[lo:hi], None if a.shadow is None else a.shadow[lo:hi],
                                      grp["lr"], self.betas[0], self.betas[1], self.eps, grp["wd"], self.sp)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        self._param_ready[key] = ev          # (several "pre" slices: the last event covers them all -- one stream)
            self._stepped = []
            return
        for gi, grp in enumerate(self.groups):
            done = sorted((lo, hi) for g_, lo, hi in self._stepped if g_ == gi)
            pos, n = 0, grp["arena"].numel
            for lo, hi in done + [(n, n)]:
                assert lo >= pos, "overlapping early-AdamW slices"
                if lo > pos:
                    self._adamw_slice(gi, pos, lo, self.step_count)
                pos = hi
        self._stepped = []

    def _step_body(self, sample_items_id, sample_items, log_mask, token_packing, defer: bool):
        self._fused_update = os.environ.get("MOREC_EARLY_ADAMW", "1") != "0"     # only here: forward_backward alone must leave the parameters untouched
        # graph mode: the dropout / DropPath kernels of THIS step fold the step block's seed word into their seeds (a captured launch keeps
        # the pointer in its arguments, so replays keep drawing fresh masks).  The registration is process-wide state of the library, so
        # it lasts exactly as long as the step: other models / steppers of the process draw the masks their own seed arguments name.
        if self.graph:
            self.sp.use_as_seed_source(True)
        try:
            try:
                loss = self.forward_backward(sample_items_id, sample_items, log_mask, token_packing)
            finally:
                self._fused_update = False
            self.reduce_gradients()
            self._in_step = defer
            try:
                self.optimizer_step()
            finally:
                self._in_step = False
        finally:
            if self.graph:
                self.sp.use_as_seed_source(False)
        return loss

    def close(self):
        """Drop the captured graphs (and with them the static inputs and the graph memory pool) and make sure the library holds no
        pointer into this stepper's step block.  The stepper stays usable: later steps run eagerly / re-capture."""
        self.flush()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._graphs.clear()
        self._graph_pool = None
        if self.sp is not None:
            self.sp.use_as_seed_source(False)

    def _throttle(self):
        """Two steps in flight keep the device queue full; the host waits for the step before the previous one (see ``step``)."""
        ev = torch.cuda.Event()
        self._inflight.append(ev)
        if len(self._inflight) > 2:
            self._inflight.pop(0).synchronize()
        return ev

    def step_graphed(self, sample_items_id, sample_items, log_mask, token_packing=None):
        """``step`` as ONE graph launch per input shape (``TrainStep(graph=True)``; otherwise plain ``step``).  The first step of a shape
        runs eagerly (allocator warm-up, first-call set-up of the kernels), the second is captured -- forward, backward, gradient
        zeroing, W^T refresh, AdamW, on both streams -- and from then on a step is: copy the batch into the graph's static input
        buffers, ``replay()``.  Per-step state the kernels need (step count, AdamW bias corrections, loss scale, dropout seed word) is
        read from the device block, so nothing of it is frozen into the graph.  Text tower: pass ``token_packing`` (host-prepared index
        vectors) or run the padded layout -- the device-side packing synchronises with the host and cannot be captured."""
        tp = token_packing
        if (not self.graph or self.dedup_items or len(self.text_attrs) > 1
                or (self.model.use_modal and not self.vision and tp is None and engine.UNPAD_DEFAULT)):
            return self.step(sample_items_id, sample_items, log_mask, token_packing)
        ins = [sample_items_id, sample_items, log_mask] + (list(tp) if tp is not None else [])
        key = tuple((tuple(t.shape), t.dtype) for t in ins)
        ent = self._graphs.get(key)
        if ent is None:            # first sight of this shape: eager
            if sum(1 for v in self._graphs.values() if v != "seen") >= self.graph_max:      # the cache is full: this shape stays eager
                return self.step(sample_items_id, sample_items, log_mask, token_packing)
            if len(self._graphs) >= 8 * self.graph_max:      # "seen" markers of shapes that never came back
                for k in [k for k, v in self._graphs.items() if v == "seen"][: len(self._graphs) // 2]:
                    del self._graphs[k]
            self._graphs[key] = "seen"
            return self.step(sample_items_id, sample_items, log_mask, token_packing)
        if ent == "seen" and sum(1 for v in self._graphs.values() if v != "seen") >= self.graph_max:
            return self.step(sample_items_id, sample_items, log_mask, token_packing)
        if ent == "seen":
            self.flush()
            static = [torch.empty_like(t) for t in ins]
            for d_, s_ in zip(static, ins):
                d_.copy_(s_)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(self.device)
            # thread_local: only THIS thread's calls are policed during capture (a collate thread may be page-locking the next batch)
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                loss = self._step_body(static[0], static[1], static[2], tuple(static[3:]) if tp is not None else None, defer=False)
            if self._graph_pool is None:
                self._graph_pool = g.pool()
            self.step_count -= 1           # (the capture pass counted a step that has not run)
            ent = self._graphs[key] = (g, static, loss)
        g, static, loss = ent
        self.flush()            # (an eager step's deferred update may still be running: the graph's forward carries no per-slice waits)
        for d_, s_ in zip(static, ins):
            d_.copy_(s_, non_blocking=True)
        ev = self._throttle()
        g.replay()
        self.step_count += 1
        ev.record(torch.cuda.current_stream(self.device))
        return loss.clone()

    def step(self, sample_items_id, sample_items, log_mask, token_packing=None):
        """The whole optimisation step of ``T/run.py:241-247`` (no GradScaler: bf16 needs no loss scaling).  Returns the loss
        of this rank's rows (device scalar); under pooled negatives that is a SHARE of the global loss -- ``global_loss``.
        ``token_packing`` (text tower, optional): the device copies of ``engine.token_packing_host(attention mask)`` uploaded with
        the batch; without it the same index vectors are derived on the device, which costs the step two host synchronisations."""
        # No synchronisation happens inside a step, so the host could issue steps far ahead of the device; every step in flight holds its
        # host batch, its events and (until its kernels are queued) its share of the allocator's attention.  Two steps in flight keep the
        # device queue full; the host waits for the step before the previous one.
        ev = self._throttle() if self.device.type == "cuda" else None
        loss = self._step_body(sample_items_id, sample_items, log_mask, token_packing, defer=True)
        if ev is not None:
            ev.record(torch.cuda.current_stream(self.device))
        return loss
