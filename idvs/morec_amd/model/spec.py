"""Parameter inventory of the hot-path model, in the reference's ``state_dict`` key order.

Key names are part of the drop-in surface: optimizer grouping matches on
``'bert_model' in name`` (``T/run.py:155``) and checkpoints store ``model.module.state_dict()``
(``T/data_utils/utils.py:109``).  Shapes follow ``T/model/modules.py:5-96`` (SASRec) and the
HuggingFace ``BertModel`` layout constructed at ``T/run.py:51-53``.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass

UE = "user_encoder.transformer_encoder."
TE = "bert_encoder.text_encoders.title."
BM = TE + "bert_model."


@dataclass
class BertShape:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12

    @staticmethod
    def named(name: str) -> "BertShape":
        """Shapes selected the way ``T/run.py:55-72`` keys on ``bert_model_load``."""
        table = {
            "tiny": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512),
            "mini": dict(hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024),
            "small": dict(hidden_size=512, num_hidden_layers=4, num_attention_heads=8, intermediate_size=2048),
            "medium": dict(hidden_size=512, num_hidden_layers=8, num_attention_heads=8, intermediate_size=2048),
            "base": dict(),
            "large": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096),
            "micro": dict(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                          intermediate_size=256, max_position_embeddings=64),
        }
        for key, kw in table.items():
            if key in name:
                return BertShape(**kw)
        raise ValueError(f"unknown bert_model_load {name!r}")


def sasrec_param_shapes(max_seq_len: int, d: int, n_blocks: int, prefix: str = UE) -> "OrderedDict[str, tuple]":
    out = OrderedDict()
    out[prefix + "position_embedding.weight"] = (max_seq_len, d)
    out[prefix + "layer_norm.weight"] = (d,)
    out[prefix + "layer_norm.bias"] = (d,)
    for l in range(n_blocks):
        a = prefix + f"transformer_blocks.{l}.multi_head_attention."
        f = prefix + f"transformer_blocks.{l}.feed_forward."
        for w in ("w_Q", "w_K", "w_V", "fc"):
            out[a + w + ".weight"] = (d, d)
        out[a + "layer_norm.weight"] = (d,)
        out[a + "layer_norm.bias"] = (d,)
        out[f + "w_1.weight"] = (4 * d, d)
        out[f + "w_1.bias"] = (4 * d,)
        out[f + "w_2.weight"] = (d, 4 * d)
        out[f + "w_2.bias"] = (d,)
        out[f + "layer_norm.weight"] = (d,)
        out[f + "layer_norm.bias"] = (d,)
    return out


def bert_param_shapes(cfg: BertShape, prefix: str = BM, pooler: bool = True) -> "OrderedDict[str, tuple]":
    H, I = cfg.hidden_size, cfg.intermediate_size
    out = OrderedDict()
    out[prefix + "embeddings.word_embeddings.weight"] = (cfg.vocab_size, H)
    out[prefix + "embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    out[prefix + "embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    out[prefix + "embeddings.LayerNorm.weight"] = (H,)
    out[prefix + "embeddings.LayerNorm.bias"] = (H,)
    for l in range(cfg.num_hidden_layers):
        L = prefix + f"encoder.layer.{l}."
        for n in ("query", "key", "value"):
            out[L + f"attention.self.{n}.weight"] = (H, H)
            out[L + f"attention.self.{n}.bias"] = (H,)
        out[L + "attention.output.dense.weight"] = (H, H)
        out[L + "attention.output.dense.bias"] = (H,)
        out[L + "attention.output.LayerNorm.weight"] = (H,)
        out[L + "attention.output.LayerNorm.bias"] = (H,)
        out[L + "intermediate.dense.weight"] = (I, H)
        out[L + "intermediate.dense.bias"] = (I,)
        out[L + "output.dense.weight"] = (H, I)
        out[L + "output.dense.bias"] = (H,)
        out[L + "output.LayerNorm.weight"] = (H,)
        out[L + "output.LayerNorm.bias"] = (H,)
    if pooler:
        out[prefix + "pooler.dense.weight"] = (H, H)
        out[prefix + "pooler.dense.bias"] = (H,)
    return out


def model_param_shapes(*, max_seq_len: int, embedding_dim: int, n_blocks: int, item_num: int, use_modal: bool,
                       bert: BertShape | None = None) -> "OrderedDict[str, tuple]":
    """Full ``Model.state_dict()`` inventory, registration order of ``T/model/model.py:16-28``."""
    out = sasrec_param_shapes(max_seq_len, embedding_dim, n_blocks)
    if use_modal:
        out.update(bert_param_shapes(bert))
        out[TE + "fc.weight"] = (embedding_dim, bert.hidden_size)
        out[TE + "fc.bias"] = (embedding_dim,)
    else:
        out["id_embedding.weight"] = (item_num + 1, embedding_dim)
    return out
