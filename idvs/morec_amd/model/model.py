"""``Model``: drop-in for ``T/model/model.py`` -- same constructor, ``forward`` signature, sub-module attribute
names (``bert_encoder`` / ``id_embedding`` / ``user_encoder``) and ``state_dict`` keys; every FLOP of the
forward AND backward runs in hand-written gfx950 kernels behind ``libmorec_hip.so``."""
import torch
import torch.distributed as dist
from torch import nn
from torch.nn.init import xavier_normal_

from .. import engine
from .. import functional as F_
from .encoders import Bert_Encoder, IdEmbedding, User_Encoder, resolve_dtype


class Model(nn.Module):
    def __init__(self, args, item_num, use_modal, bert_model, pop_prob_list):
        super().__init__()
        self.args = args
        self.use_modal = use_modal
        self.max_seq_len = args.max_seq_len
        self.compute_dtype = resolve_dtype(args)
        self.pop_prob_list = torch.FloatTensor(pop_prob_list)        # plain attribute, as in T/model/model.py:14
        self._log_pop = None                                          # log(pop) table, built once per device
        # pooled negatives across ranks; off = the reference's rank-local negatives
        self.pool_negatives = bool(getattr(args, "pool_negatives", False))
        # local loss share is multiplied by this (world_size when a gradient-averaging DDP wrapper follows)
        self.pool_loss_mult = None
        self.user_encoder = User_Encoder(item_num=item_num, max_seq_len=args.max_seq_len, item_dim=args.embedding_dim,
                                         num_attention_heads=args.num_attention_heads, dropout=args.drop_rate,
                                         n_layers=args.transformer_block, compute_dtype=self.compute_dtype)
        if self.use_modal:
            self.bert_encoder = Bert_Encoder(args=args, bert_model=bert_model)
        else:
            self.id_embedding = IdEmbedding(item_num + 1, args.embedding_dim, padding_idx=0,
                                            compute_dtype=self.compute_dtype)
            xavier_normal_(self.id_embedding.weight.data)            # T/model/model.py:28 (overwrites the pad row too)
        self.criterion = nn.CrossEntropyLoss()                       # kept for attribute compatibility; unused

    def _log_pop_table(self, device):
        if self._log_pop is None or self._log_pop.device != device:
            self._log_pop = torch.log(self.pop_prob_list).to(device)  # T/model/model.py:33, hoisted out of the step
        return self._log_pop

    def forward(self, sample_items_id, sample_items, log_mask, local_rank=None):
        if float(getattr(self.args, "drop_rate", 0.0)) > 0 and self.training and not getattr(self.args, "allow_no_dropout", False):
            raise NotImplementedError("training-mode dropout is not implemented in the HIP path yet: use model.eval(), "
                                      "drop_rate=0 or args.allow_no_dropout=True")
        D = self.args.embedding_dim
        ids = sample_items_id.view(-1)
        if self.use_modal:
            score_embs = self.bert_encoder.encode(sample_items)
        else:
            score_embs = self.id_embedding.encode(sample_items.view(-1))
        input_embs = score_embs.view(-1, self.max_seq_len + 1, D)
        prec_vec = self.user_encoder.encode(input_embs[:, :-1, :], log_mask).reshape(-1, D)
        ci = engine.ce_inputs_local(ids, log_mask, self._log_pop_table(ids.device))
        pooled = self.pool_negatives and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        mult = self.pool_loss_mult if self.pool_loss_mult is not None else (float(dist.get_world_size()) if pooled else 1.0)
        return F_.InBatchCEFn.apply(prec_vec, score_embs, ci, pooled, mult, engine)
