"""``Model``: drop-in for ``T/model/model.py`` and ``V/model/model.py`` -- same constructor, ``forward`` signature,
sub-module attribute names (``bert_encoder`` | ``cv_encoder`` / ``id_embedding`` / ``user_encoder``) and ``state_dict``
keys; every FLOP of the forward AND backward runs in hand-written gfx950 kernels behind ``libmorec_hip.so``.  The
vision variant is selected the way the reference's two packages differ: ``args.CV_model_load`` is present
(``V/model/model.py:24-29``) and the fourth constructor argument is the image network."""
import torch
import torch.distributed as dist
from torch import nn
from torch.nn.init import xavier_normal_

from .. import engine, ops
from .. import functional as F_
from .encoders import Bert_Encoder, IdEmbedding, User_Encoder, Vit_Encoder, resolve_dtype, resolve_fp32_gemm, resolve_res32


class Model(nn.Module):
    def __init__(self, args, item_num, use_modal, bert_model, pop_prob_list):
        super().__init__()
        self.args = args
        self.use_modal = use_modal
        self.max_seq_len = args.max_seq_len
        self.compute_dtype = resolve_dtype(args)
        self.fp32_gemm = resolve_fp32_gemm(args)                      # "exact" | "bf16x3": how fp32 GEMMs run (ops.FP32_GEMM)
        self.res32 = resolve_res32(args) and not hasattr(args, "CV_model_load")      # fp32 residual stream (autocast data flow): text / ID towers
        if resolve_res32(args) and not self.res32:
            import sys
            print(f"morec: compute_dtype={getattr(args, 'compute_dtype', None)!r}: the vision tower has no fp32-residual-stream kernels (pre-LN Swin blocks keep "
                  f"their 16-bit stream); running it as {str(self.compute_dtype).replace('torch.', '')!r}", file=sys.stderr)
        self.pop_prob_list = torch.FloatTensor(pop_prob_list)        # plain attribute, as in T/model/model.py:14
        self._log_pop = None                                          # log(pop) table, built once per device
        # pooled negatives across ranks; off = the reference's rank-local negatives
        self.pool_negatives = bool(getattr(args, "pool_negatives", False))
        # local loss share is multiplied by this (world_size when a gradient-averaging DDP wrapper follows)
        self.pool_loss_mult = None
        self.user_encoder = User_Encoder(item_num=item_num, max_seq_len=args.max_seq_len, item_dim=args.embedding_dim,
                                         num_attention_heads=args.num_attention_heads, dropout=args.drop_rate,
                                         n_layers=args.transformer_block, compute_dtype=self.compute_dtype, res32=self.res32)
        self.vision = hasattr(args, "CV_model_load")
        if self.use_modal and self.vision:
            if "swin" not in args.CV_model_load:
                raise NotImplementedError("only the Swin tower of V/model/model.py:24-29 is implemented (ResNet / BEiT / MAE "
                                          "launchers are outside the benchmarked path)")
            self.cv_encoder = Vit_Encoder(image_net=bert_model, compute_dtype=self.compute_dtype)
        elif self.use_modal:
            self.bert_encoder = Bert_Encoder(args=args, bert_model=bert_model)
        else:
            self.id_embedding = IdEmbedding(item_num + 1, args.embedding_dim, padding_idx=0,
                                            compute_dtype=self.compute_dtype)
            xavier_normal_(self.id_embedding.weight.data)            # T/model/model.py:28 (overwrites the pad row too)
        self.criterion = nn.CrossEntropyLoss()                       # kept for attribute compatibility; unused
        self._drop_calls = 0

    def dropout_cfgs(self):
        """Per-call dropout streams (training mode only).  SASRec uses ``args.drop_rate`` on hidden states and attention
        probabilities (``T/model/modules.py:9,24,48``); BERT uses its own config's probabilities (HF)."""
        if not self.training:
            return engine.NO_DROP, engine.NO_DROP
        self._drop_calls += 1
        base = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._drop_calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        if dist.is_available() and dist.is_initialized():
            base = (base + dist.get_rank() * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        p = float(self.args.drop_rate)
        d_user = engine.DropCfg(p, p, base ^ 0x5555555555555555) if p > 0 else engine.NO_DROP
        d_item = engine.NO_DROP
        if self.use_modal and self.vision:
            d_item = engine.DropCfg(0.0, 0.0, base)      # only the DropPath streams are drawn from it
        elif self.use_modal:
            c = self.bert_encoder.text_encoders["title"].bert_model.config
            ph, pa = float(getattr(c, "hidden_dropout_prob", 0.0)), float(getattr(c, "attention_probs_dropout_prob", 0.0))
            if ph > 0 or pa > 0:
                d_item = engine.DropCfg(ph, pa, base)
        return d_item, d_user

    def _log_pop_table(self, device):
        if self._log_pop is None or self._log_pop.device != device:
            self._log_pop = torch.log(self.pop_prob_list).to(device)  # T/model/model.py:33, hoisted out of the step
        return self._log_pop

    def forward(self, sample_items_id, sample_items, log_mask, local_rank=None):
        with ops.fp32_gemm_mode(self.fp32_gemm):      # "exact" | "bf16x3"; the autograd shells carry it into their backward
            return self._forward(sample_items_id, sample_items, log_mask, local_rank)

    def _forward(self, sample_items_id, sample_items, log_mask, local_rank=None):
        D = self.args.embedding_dim
        ids = sample_items_id.view(-1)
        d_item, d_user = self.dropout_cfgs()
        if self.use_modal and self.vision:
            score_embs = self.cv_encoder.encode(sample_items, d_item)
        elif self.use_modal:
            score_embs = self.bert_encoder.encode(sample_items, d_item)
        else:
            score_embs = self.id_embedding.encode(sample_items.view(-1))
        input_embs = score_embs.view(-1, self.max_seq_len + 1, D)
        prec_vec = self.user_encoder.encode(input_embs[:, :-1, :], log_mask, d_user).reshape(-1, D)
        ci = engine.ce_inputs_local(ids, log_mask, self._log_pop_table(ids.device))
        pooled = self.pool_negatives and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        mult = self.pool_loss_mult if self.pool_loss_mult is not None else (float(dist.get_world_size()) if pooled else 1.0)
        return F_.InBatchCEFn.apply(prec_vec, score_embs, ci, pooled, mult, engine)
