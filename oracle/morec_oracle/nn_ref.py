"""Floating-point part of the hot path -- PyTorch-CPU fp32 restatement.

TEST INFRASTRUCTURE ONLY (see package docstring).  Forward passes are written out op by op;
gradients come from torch autograd over these forwards.  Parameters are passed as dicts keyed by
the reference's ``state_dict`` names so goldens, the oracle and the HIP model share one naming.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import bookkeeping as bk


def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# --------------------------------------------------------------------------------------
# SASRec user encoder
# --------------------------------------------------------------------------------------
def sasrec_forward(p: dict, input_embs: torch.Tensor, log_mask: torch.Tensor, n_heads: int,
                   prefix: str = "user_encoder.transformer_encoder.") -> torch.Tensor:
    """``User_Encoder.forward`` ``T/model/encoders.py:23-28`` + ``TransformerEncoder.forward``
    ``T/model/modules.py:89-96`` (dropout off): additive mask 0 / -1e9 from
    ``tril(log_mask != 0)``; ``x = LN_1e-6(in + pos)``; per block post-LN MHA without biases
    (``modules.py:52-63``, softmax(QK^T/sqrt(d_k) + mask)) and post-LN FFN D->4D->D ReLU
    (``modules.py:14-17``).  input_embs [B,S,D], log_mask [B,S] -> [B,S,D]."""
    B, S, D = input_embs.shape
    dk = D // n_heads
    keep = (log_mask != 0)[:, None, None, :].expand(B, 1, S, S)
    keep = torch.tril(keep)
    att_mask = torch.where(keep, torch.tensor(0.0), torch.tensor(-1e9))
    x = _ln(input_embs + p[prefix + "position_embedding.weight"][:S][None], p[prefix + "layer_norm.weight"],
            p[prefix + "layer_norm.bias"], 1e-6)
    l = 0
    while (prefix + f"transformer_blocks.{l}.multi_head_attention.w_Q.weight") in p:
        a = prefix + f"transformer_blocks.{l}.multi_head_attention."
        f = prefix + f"transformer_blocks.{l}.feed_forward."
        q = (x @ p[a + "w_Q.weight"].t()).view(B, S, n_heads, dk).transpose(1, 2)
        k = (x @ p[a + "w_K.weight"].t()).view(B, S, n_heads, dk).transpose(1, 2)
        v = (x @ p[a + "w_V.weight"].t()).view(B, S, n_heads, dk).transpose(1, 2)
        att = q @ k.transpose(-2, -1) / (dk ** 0.5) + att_mask
        ctx = (torch.softmax(att, dim=-1) @ v).transpose(1, 2).contiguous().view(B, S, D)
        x = _ln(x + ctx @ p[a + "fc.weight"].t(), p[a + "layer_norm.weight"], p[a + "layer_norm.bias"], 1e-6)
        h = torch.relu(x @ p[f + "w_1.weight"].t() + p[f + "w_1.bias"])
        x = _ln(x + h @ p[f + "w_2.weight"].t() + p[f + "w_2.bias"], p[f + "layer_norm.weight"],
                p[f + "layer_norm.bias"], 1e-6)
        l += 1
    return x


# --------------------------------------------------------------------------------------
# BERT item encoder (third-party arithmetic: HuggingFace transformers, eager attention)
# --------------------------------------------------------------------------------------
def bert_forward(p: dict, input_ids: torch.Tensor, attention_mask: torch.Tensor, n_heads: int,
                 prefix: str = "", eps: float = 1e-12, mask_value: float | None = None) -> torch.Tensor:
    """``BertModel.forward(...)[0]`` as called at ``T/model/encoders.py:68``.  Published algorithm
    (HF ``models/bert/modeling_bert.py``): embeddings = word + position(0..T-1) + token_type(0)
    -> LN(eps 1e-12); per layer: Q/K/V Linear with bias, softmax(QK^T/sqrt(dh) + additive key
    mask), PV, dense + residual + LN (post-LN), dense + erf-GELU, dense + residual + LN.
    ``mask_value``: additive value on masked keys -- ``finfo(float32).min`` is what transformers
    5.x eager applies (default here); transformers 4.20.1 used -10000."""
    if mask_value is None:
        mask_value = torch.finfo(torch.float32).min
    N, T = input_ids.shape
    H = p[prefix + "embeddings.word_embeddings.weight"].shape[1]
    dh = H // n_heads
    # nn.Embedding(vocab, H, padding_idx=pad_token_id=0): the [PAD] row receives no gradient
    x = (F.embedding(input_ids, p[prefix + "embeddings.word_embeddings.weight"], padding_idx=0)
         + p[prefix + "embeddings.position_embeddings.weight"][:T][None]
         + p[prefix + "embeddings.token_type_embeddings.weight"][0][None, None])
    x = _ln(x, p[prefix + "embeddings.LayerNorm.weight"], p[prefix + "embeddings.LayerNorm.bias"], eps)
    add_mask = torch.where(attention_mask[:, None, None, :] != 0, torch.tensor(0.0), torch.tensor(mask_value))
    l = 0
    while (prefix + f"encoder.layer.{l}.attention.self.query.weight") in p:
        L = prefix + f"encoder.layer.{l}."
        q = (x @ p[L + "attention.self.query.weight"].t() + p[L + "attention.self.query.bias"]).view(N, T, n_heads, dh).transpose(1, 2)
        k = (x @ p[L + "attention.self.key.weight"].t() + p[L + "attention.self.key.bias"]).view(N, T, n_heads, dh).transpose(1, 2)
        v = (x @ p[L + "attention.self.value.weight"].t() + p[L + "attention.self.value.bias"]).view(N, T, n_heads, dh).transpose(1, 2)
        att = q @ k.transpose(-2, -1) * (dh ** -0.5) + add_mask
        ctx = (torch.softmax(att, dim=-1) @ v).transpose(1, 2).contiguous().view(N, T, H)
        x = _ln(x + ctx @ p[L + "attention.output.dense.weight"].t() + p[L + "attention.output.dense.bias"],
                p[L + "attention.output.LayerNorm.weight"], p[L + "attention.output.LayerNorm.bias"], eps)
        h = _gelu_erf(x @ p[L + "intermediate.dense.weight"].t() + p[L + "intermediate.dense.bias"])
        x = _ln(x + h @ p[L + "output.dense.weight"].t() + p[L + "output.dense.bias"],
                p[L + "output.LayerNorm.weight"], p[L + "output.LayerNorm.bias"], eps)
        l += 1
    return x


def text_encoder_forward(p: dict, text: torch.Tensor, n_heads: int,
                         prefix: str = "bert_encoder.text_encoders.title.") -> torch.Tensor:
    """``Text_Encoder.forward`` ``T/model/encoders.py:63-70``: split [ids | attention_mask] down the
    middle, BERT, ``GELU(fc(hidden[:, 0]))``.  text int64[Nc, 2T] -> [Nc, D]."""
    T = text.shape[1] // 2
    ids, mask = text[:, :T], text[:, T:]
    hidden = bert_forward(p, ids, mask, n_heads, prefix=prefix + "bert_model.")
    cls = hidden[:, 0] @ p[prefix + "fc.weight"].t() + p[prefix + "fc.bias"]
    return _gelu_erf(cls)


# --------------------------------------------------------------------------------------
# In-batch debiased cross-entropy
# --------------------------------------------------------------------------------------
def inbatch_ce_loss(prec_vec: torch.Tensor, score_embs: torch.Tensor, sample_items_id, log_mask,
                    pop_prob_list, max_seq_len: int, *, pool_ids=None, pool_log_mask=None,
                    col_offset: int = 0, n_valid_total: int | None = None, return_parts: bool = False):
    """``T/model/model.py:32-33,45-67`` vectorised: logits = P @ E^T - log(pop[ids]); invalid
    columns and rejected cells := -1e4 (positive restored); mean CE over rows with log_mask != 0.

    Pooled form: ``score_embs``/``pool_ids``/``pool_log_mask`` describe all ranks'
    columns (rank-major), the local positives start at ``col_offset``, and the sum of row losses
    is divided by ``n_valid_total`` (global count) instead of the local count.
    """
    S = max_seq_len
    log_mask_np = np.asarray(log_mask.detach().cpu() if torch.is_tensor(log_mask) else log_mask, dtype=np.float32)
    bs = log_mask_np.shape[0]
    ids_np = np.asarray(sample_items_id.detach().cpu() if torch.is_tensor(sample_items_id) else sample_items_id).reshape(-1)
    cols_np = ids_np if pool_ids is None else np.asarray(pool_ids).reshape(-1)
    cols_mask_np = log_mask_np if pool_log_mask is None else np.asarray(pool_log_mask, dtype=np.float32)
    debias = torch.from_numpy(bk.log_pop(pop_prob_list, cols_np))
    logits = prec_vec @ score_embs.t() - debias[None, :]
    colvalid = torch.from_numpy(bk.column_valid(cols_mask_np))
    rej = torch.from_numpy(bk.reject_mask(ids_np, bs, S, pool_ids=cols_np, col_offset=col_offset)).view(bs * S, -1)
    masked = (~colvalid)[None, :] | rej
    logits = torch.where(masked, torch.tensor(-1e4), logits)
    rows = torch.from_numpy(bk.valid_rows(log_mask_np))
    labels = torch.from_numpy(bk.ce_labels(bs, S) + col_offset)
    lsm = torch.log_softmax(logits[rows], dim=-1)
    row_loss = -lsm[torch.arange(rows.numel()), labels[rows]]
    n = rows.numel() if n_valid_total is None else n_valid_total
    loss = row_loss.sum() / n
    if return_parts:
        return loss, dict(masked=masked, rows=rows, labels=labels, logits=logits)
    return loss


def model_forward(p: dict, sample_items_id, sample_items, log_mask, pop_prob_list, *, max_seq_len: int,
                  embedding_dim: int, n_heads: int, use_modal: bool, bert_heads: int = 12, item_vecs=None):
    """``Model.forward`` ``T/model/model.py:31-69`` (single process, dropout off).  ``item_vecs``: already encoded item
    vectors (the vision tower, ``V/model/model.py:38-39``, is restated in ``swin_ref``)."""
    if item_vecs is not None:
        score_embs = item_vecs
    elif use_modal:
        score_embs = text_encoder_forward(p, sample_items, bert_heads)
    else:
        score_embs = p["id_embedding.weight"][sample_items]
    input_embs = score_embs.view(-1, max_seq_len + 1, embedding_dim)
    prec = sasrec_forward(p, input_embs[:, :-1, :], log_mask, n_heads).reshape(-1, embedding_dim)
    return inbatch_ce_loss(prec, score_embs, sample_items_id, log_mask, pop_prob_list, max_seq_len)
