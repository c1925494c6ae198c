"""BCE variant of the loss -- PyTorch-CPU fp32 restatement.  TEST INFRASTRUCTURE ONLY (see package docstring).
Follows ``bce_text/main-end2end/model/model.py:30-51``; pinned by ``tests/golden/g14_bce.npz``."""
import torch
import torch.nn.functional as F

from .nn_ref import sasrec_forward, text_encoder_forward


def bce_loss(prec: torch.Tensor, emb: torch.Tensor, log_mask: torch.Tensor) -> torch.Tensor:
    """prec [B, S, D], emb [B, S+1, 2, D] (pos, neg), log_mask [B, S]: two mean BCE-with-logits terms over the rows with
    ``log_mask != 0`` (model.py:42-50): targets 1 for (prec . pos[j+1]), 0 for (prec . neg[j])."""
    pos = (prec * emb[:, 1:, 0]).sum(-1)
    neg = (prec * emb[:, :-1, 1]).sum(-1)
    idx = torch.where(log_mask != 0)
    return F.softplus(-pos[idx]).mean() + F.softplus(neg[idx]).mean()


def bce_model_forward(p: dict, sample_items, log_mask, *, max_seq_len: int, embedding_dim: int, n_heads: int, use_modal: bool,
                      bert_heads: int = 12):
    """``Model.forward`` of the BCE variant (model.py:30-51), dropout off."""
    if use_modal:
        E = text_encoder_forward(p, sample_items.reshape(-1, sample_items.shape[-1]), bert_heads)
    else:
        E = p["id_embedding.weight"][sample_items.reshape(-1)]
    emb = E.view(-1, max_seq_len + 1, 2, embedding_dim)
    prec = sasrec_forward(p, emb[:, :-1, 0], log_mask, n_heads)
    return bce_loss(prec, emb, log_mask)
