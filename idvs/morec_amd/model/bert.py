"""``HipBertModel``: parameter container with HuggingFace ``BertModel``'s module tree and ``state_dict`` keys
(the reference builds ``BertModel.from_pretrained(...)`` at ``T/run.py:51-53`` and reaches it as
``model.module.bert_encoder.text_encoders.title.bert_model``, ``T/run.py:165``).  Forward arithmetic is in
``libmorec_hip.so`` (``functional.BertEncoderFn``); this class only owns the weights, so a HF checkpoint's
``state_dict`` loads key for key (``from_hf``)."""
import types

import torch
import torch.nn as nn

from .spec import BertShape


class _Self(nn.Module):
    def __init__(self, H):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)


class _SelfOutput(nn.Module):
    def __init__(self, i, o, eps):
        super().__init__()
        self.dense = nn.Linear(i, o)
        self.LayerNorm = nn.LayerNorm(o, eps=eps)


class _Attention(nn.Module):
    def __init__(self, H, eps):
        super().__init__()
        self.self = _Self(H)
        self.output = _SelfOutput(H, H, eps)


class _Intermediate(nn.Module):
    def __init__(self, H, I):
        super().__init__()
        self.dense = nn.Linear(H, I)


class _Layer(nn.Module):
    def __init__(self, H, I, eps):
        super().__init__()
        self.attention = _Attention(H, eps)
        self.intermediate = _Intermediate(H, I)
        self.output = _SelfOutput(I, H, eps)


class _Encoder(nn.Module):
    def __init__(self, L, H, I, eps):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(H, I, eps) for _ in range(L)])


class _Embeddings(nn.Module):
    def __init__(self, c: BertShape):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _Pooler(nn.Module):
    def __init__(self, H):
        super().__init__()
        self.dense = nn.Linear(H, H)


class HipBertModel(nn.Module):
    def __init__(self, shape: BertShape, initializer_range: float = 0.02, hidden_dropout_prob: float = 0.1,
                 attention_probs_dropout_prob: float = 0.1):
        super().__init__()
        self.shape = shape
        self.config = types.SimpleNamespace(hidden_dropout_prob=hidden_dropout_prob,
                                            attention_probs_dropout_prob=attention_probs_dropout_prob,
                                            hidden_size=shape.hidden_size, num_hidden_layers=shape.num_hidden_layers,
                                            num_attention_heads=shape.num_attention_heads,
                                            intermediate_size=shape.intermediate_size, vocab_size=shape.vocab_size,
                                            max_position_embeddings=shape.max_position_embeddings,
                                            layer_norm_eps=shape.layer_norm_eps, pad_token_id=0)
        self.embeddings = _Embeddings(shape)
        self.encoder = _Encoder(shape.num_hidden_layers, shape.hidden_size, shape.intermediate_size, shape.layer_norm_eps)
        self.pooler = _Pooler(shape.hidden_size)   # present for state_dict compatibility; never evaluated (its
        #                                            output is discarded by the reference, T/model/encoders.py:68-69)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=initializer_range)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=initializer_range)
                if m.padding_idx is not None:
                    with torch.no_grad():
                        m.weight[m.padding_idx].zero_()

    @staticmethod
    def from_hf(hf_model) -> "HipBertModel":
        """Adopt a HuggingFace ``BertModel`` (what ``T/run.py:53`` constructs): same keys, weights copied."""
        c = hf_model.config
        shape = BertShape(vocab_size=c.vocab_size, hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers,
                          num_attention_heads=c.num_attention_heads, intermediate_size=c.intermediate_size,
                          max_position_embeddings=c.max_position_embeddings, type_vocab_size=c.type_vocab_size,
                          layer_norm_eps=c.layer_norm_eps)
        m = HipBertModel(shape, hidden_dropout_prob=getattr(c, "hidden_dropout_prob", 0.1),
                         attention_probs_dropout_prob=getattr(c, "attention_probs_dropout_prob", 0.1))
        missing, unexpected = m.load_state_dict(hf_model.state_dict(), strict=False)
        assert not missing, missing
        for (n1, p1), (n2, p2) in zip(m.named_parameters(), hf_model.named_parameters()):
            assert n1 == n2, (n1, n2)       # same ORDER too: T/run.py:73-75 freezes parameters by index
            p1.requires_grad = p2.requires_grad
        return m
