"""Parameter containers of the SASRec user encoder with the reference's module tree and ``state_dict`` keys
(``T/model/modules.py:5-96``).  The ``nn.Linear`` / ``nn.LayerNorm`` / ``nn.Embedding`` objects only OWN the
parameters (same names, shapes and initialisation as the reference); their ``forward`` is never called --
the arithmetic runs in ``libmorec_hip.so`` through ``functional.SasrecFn``."""
import torch.nn as nn


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model, d_inner, dropout):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_inner)
        self.w_2 = nn.Linear(d_inner, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.dropout_p = dropout


class MultiHeadedAttention(nn.Module):
    def __init__(self, n_heads, d_model, dropout):
        super().__init__()
        assert d_model % n_heads == 0
        self.d_model, self.n_heads, self.d_k = d_model, n_heads, d_model // n_heads
        self.w_Q = nn.Linear(d_model, d_model, bias=False)
        self.w_K = nn.Linear(d_model, d_model, bias=False)
        self.w_V = nn.Linear(d_model, d_model, bias=False)
        self.fc = nn.Linear(d_model, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.dropout_p = dropout


class TransformerBlock(nn.Module):
    def __init__(self, d_model, n_heads, d_inner, dropout):
        super().__init__()
        self.multi_head_attention = MultiHeadedAttention(n_heads=n_heads, d_model=d_model, dropout=dropout)
        self.feed_forward = PositionwiseFeedForward(d_model=d_model, d_inner=d_inner, dropout=dropout)


class TransformerEncoder(nn.Module):
    def __init__(self, n_vocab, n_position, d_model, n_heads, dropout, n_layers):
        super().__init__()
        self.position_embedding = nn.Embedding(n_position, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.transformer_blocks = nn.ModuleList(
            [TransformerBlock(d_model=d_model, n_heads=n_heads, d_inner=d_model * 4, dropout=dropout)
             for _ in range(n_layers)])
        self.n_heads, self.n_layers, self.d_model, self.dropout_p = n_heads, n_layers, d_model, dropout
