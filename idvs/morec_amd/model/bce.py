"""``BceModel``: drop-in for ``bce_text/main-end2end/model/model.py`` -- the same encoders as the in-batch
``Model`` with the one-sampled-negative BCE loss of ``model.py:30-51``: constructor ``(args, item_num, use_modal,
bert_model)``, ``forward(sample_items, log_mask, local_rank)``, attributes ``user_encoder`` / ``bert_encoder`` /
``id_embedding``, identical ``state_dict`` keys.  Scoring, loss and their backward are ``morec_bce_fwd/bwd``."""
import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from .encoders import Bert_Encoder, IdEmbedding, User_Encoder, resolve_dtype, resolve_fp32_gemm, resolve_res32


class BceLossFn(torch.autograd.Function):
    """P [B, S, D] user states, E [B*(S+1)*2, D] item vectors (pos / neg interleaved), log_mask [B, S] -> scalar loss."""

    @staticmethod
    def forward(ctx, P, E, log_mask):
        B, S = log_mask.shape
        P, E = P.contiguous(), E.contiguous()
        row_valid = (log_mask.reshape(-1) != 0).to(torch.uint8).contiguous()
        n_valid = row_valid.sum(dtype=torch.float32)
        loss_sum, scores = ops.bce_fwd(P, E, row_valid, B, S)
        ctx.stuff = (P, E, row_valid, scores, n_valid, B, S)
        return (loss_sum[0] / n_valid).to(torch.float32)

    @staticmethod
    def backward(ctx, dloss):
        P, E, row_valid, scores, n_valid, B, S = ctx.stuff
        ctx.stuff = None
        g = (dloss.to(torch.float32) / n_valid).reshape(1).contiguous()
        dP, dE = ops.bce_bwd(P, E, row_valid, scores, g, B, S)
        return dP, dE, None


class BceModel(nn.Module):
    def __init__(self, args, item_num, use_modal, bert_model):
        super().__init__()
        self.args = args
        self.use_modal = use_modal
        self.max_seq_len = args.max_seq_len + 1          # bce model.py:13
        self.compute_dtype = resolve_dtype(args)
        self.fp32_gemm = resolve_fp32_gemm(args)          # "exact" | "bf16x3": how fp32 GEMMs run for this model (ops.FP32_GEMM)
        self.user_encoder = User_Encoder(item_num=item_num, max_seq_len=args.max_seq_len, item_dim=args.embedding_dim,
                                         num_attention_heads=args.num_attention_heads, dropout=args.drop_rate,
                                         n_layers=args.transformer_block, compute_dtype=self.compute_dtype, res32=resolve_res32(args))
        if self.use_modal:
            self.bert_encoder = Bert_Encoder(args=args, bert_model=bert_model)
        else:
            self.id_embedding = IdEmbedding(item_num + 1, args.embedding_dim, padding_idx=0, compute_dtype=self.compute_dtype)
            xavier_normal_(self.id_embedding.weight.data)
        self.criterion = nn.BCEWithLogitsLoss()          # attribute compatibility; unused

    def forward(self, sample_items, log_mask, local_rank=None):
        with ops.fp32_gemm_mode(self.fp32_gemm):      # the autograd shells carry the mode into their backward
            return self._forward(sample_items, log_mask, local_rank)

    def _forward(self, sample_items, log_mask, local_rank=None):
        D = self.args.embedding_dim
        if self.use_modal:
            E = self.bert_encoder.encode(sample_items.reshape(-1, sample_items.shape[-1]))
        else:
            E = self.id_embedding.encode(sample_items.reshape(-1))
        E = E.reshape(-1, D)
        emb = E.view(-1, self.max_seq_len, 2, D)
        P = self.user_encoder.encode(emb[:, :-1, 0].contiguous(), log_mask)     # positives of slots 0..S-1 (model.py:40)
        return BceLossFn.apply(P, E, log_mask)
