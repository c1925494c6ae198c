"""Swin vision item encoder -- PyTorch-CPU fp32 restatement.  TEST INFRASTRUCTURE ONLY (see package docstring).

Path: ``Vit_Encoder.forward`` ``V/model/encoders.py:24-31`` = ``GELU(image_net(x)[0])`` where ``image_net`` is the
third-party HuggingFace ``SwinForImageClassification`` built at ``V/run.py:47-54`` (classifier replaced by
``Linear(num_features, embedding_dim)``).  The Swin arithmetic lives in ``transformers`` (reference pins 4.20.1,
``README.md:44``; this container has 5.15.0): ``transformers/models/swin/modeling_swin.py`` -- patch embedding
``:247-286`` + LN ``:167-246``, ``SwinLayer`` ``:508-626`` (pre-LN, window partition ``:486-505``, cyclic shift with the
-100 region mask ``:584-607``, relative-position bias ``:329-370``, q/k/v/o ``:401-468``, DropPath on the attention branch
``:42-60``), ``SwinPatchMerging`` ``:289-326``, final LN + mean pool ``:876-881``, classifier ``:1048-1050``.
Parameters are a dict keyed by the installed-HF ``state_dict`` names (prefix ``cv_encoder.image_net.``).
Pinned by ``tests/golden/g11_swin_micro.npz`` (captured from the imported reference + installed HF).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class SwinCfg:
    image_size: int = 224
    patch_size: int = 4
    num_channels: int = 3
    embed_dim: int = 96
    depths: tuple = (2, 2, 6, 2)
    num_heads: tuple = (3, 6, 12, 24)
    window_size: int = 7
    mlp_ratio: float = 4.0
    layer_norm_eps: float = 1e-5
    drop_path_rate: float = 0.1
    extra: dict = field(default_factory=dict)


def rel_position_index(ws: int) -> torch.Tensor:
    """[ws*ws, ws*ws] index into the (2ws-1)^2-row bias table (``modeling_swin.py:350-365``)."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)   # [2, T]
    rel = c[:, :, None] - c[:, None, :]
    return (rel[0] + ws - 1) * (2 * ws - 1) + (rel[1] + ws - 1)


def shift_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor | None:
    """[n_windows, T, T] additive 0 / -100 mask between tokens of different cyclic-shift regions (``:584-607``)."""
    if shift <= 0:
        return None
    hi, wi = torch.arange(H), torch.arange(W)
    hr = (hi >= H - ws).long() + (hi >= H - shift).long()
    wr = (wi >= W - ws).long() + (wi >= W - shift).long()
    img = (hr[:, None] * 3 + wr[None, :]).float()                                     # [H, W]
    win = img.view(H // ws, ws, W // ws, ws).transpose(1, 2).reshape(-1, ws * ws)     # window_partition
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


def drop_path_rates(cfg: SwinCfg):
    n = sum(cfg.depths)
    return [cfg.drop_path_rate * i / max(n - 1, 1) for i in range(n)]      # ``:758``


def _partition(x, ws):   # [B, H, W, C] -> [B*nW, ws*ws, C]
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).transpose(2, 3).reshape(-1, ws * ws, C)


def _reverse(w, ws, H, W):   # inverse of _partition
    C = w.shape[-1]
    return w.view(-1, H // ws, W // ws, ws, ws, C).transpose(2, 3).reshape(-1, H, W, C)


def swin_layer(p, L, x, H, W, heads, ws, shift, eps, keep_scale=None):
    """One ``SwinLayer`` (``:508-572``).  x [B, H*W, C]; keep_scale [B] = DropPath keep / keep_prob (None = eval)."""
    B, _, C = x.shape
    dh = C // heads
    if min(H, W) <= ws:          # ``:574-581``
        shift, ws = 0, min(H, W)
    assert H % ws == 0 and W % ws == 0, "padding to a window multiple is outside the hot-path configurations"
    h = F.layer_norm(x, (C,), p[L + "layernorm_before.weight"], p[L + "layernorm_before.bias"], eps).view(B, H, W, C)
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    win = _partition(h, ws)                                                       # [B*nW, T, C]
    T = ws * ws
    A = L + "attention."
    q = (win @ p[A + "q_proj.weight"].t() + p[A + "q_proj.bias"]).view(-1, T, heads, dh).transpose(1, 2)
    k = (win @ p[A + "k_proj.weight"].t() + p[A + "k_proj.bias"]).view(-1, T, heads, dh).transpose(1, 2)
    v = (win @ p[A + "v_proj.weight"].t() + p[A + "v_proj.bias"]).view(-1, T, heads, dh).transpose(1, 2)
    table = p[A + "relative_position_bias.relative_position_bias_table"]
    bias = table[rel_position_index(ws).view(-1)].view(T, T, heads).permute(2, 0, 1)[None]      # [1, heads, T, T]
    s = q @ k.transpose(-2, -1) * (dh ** -0.5) + bias
    m = shift_mask(H, W, ws, shift)
    if m is not None:
        nW = m.shape[0]
        s = (s.view(B, nW, heads, T, T) + m[None, :, None]).view(-1, heads, T, T)
    ctx = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(-1, T, C)
    a = ctx @ p[A + "o_proj.weight"].t() + p[A + "o_proj.bias"]
    a = _reverse(a, ws, H, W)
    if shift > 0:
        a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
    a = a.reshape(B, H * W, C)
    if keep_scale is not None:
        a = a * keep_scale.view(B, 1, 1)
    x = x + a
    y = F.layer_norm(x, (C,), p[L + "layernorm_after.weight"], p[L + "layernorm_after.bias"], eps)
    y = F.gelu(y @ p[L + "mlp.fc1.weight"].t() + p[L + "mlp.fc1.bias"])
    return x + y @ p[L + "mlp.fc2.weight"].t() + p[L + "mlp.fc2.bias"]


def patch_merge(p, Dn, x, H, W):
    """``SwinPatchMerging`` (``:309-326``): channel blocks ordered (row0,col0), (row1,col0), (row0,col1), (row1,col1)."""
    B, _, C = x.shape
    assert H % 2 == 0 and W % 2 == 0
    g = x.view(B, H, W, C)
    g = torch.cat([g[:, r::2, c::2, :] for c in range(2) for r in range(2)], dim=-1).reshape(B, -1, 4 * C)
    g = F.layer_norm(g, (4 * C,), p[Dn + "norm.weight"], p[Dn + "norm.bias"], 1e-5)   # nn.LayerNorm default eps
    return g @ p[Dn + "reduction.weight"].t()


def swin_forward(p: dict, cfg: SwinCfg, pixels: torch.Tensor, prefix: str = "cv_encoder.image_net.",
                 keep_scales=None) -> torch.Tensor:
    """``SwinForImageClassification.forward(pixels)[0]``: pixels [N, 3, R, R] -> logits [N, num_labels].
    keep_scales: list (one per layer) of [N] DropPath scales, or None for eval mode."""
    sw = prefix + "swin."
    N = pixels.shape[0]
    ps = cfg.patch_size
    x = F.conv2d(pixels, p[sw + "embeddings.patch_embeddings.projection.weight"],
                 p[sw + "embeddings.patch_embeddings.projection.bias"], stride=ps)
    H, W = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    C = cfg.embed_dim
    x = F.layer_norm(x, (C,), p[sw + "embeddings.norm.weight"], p[sw + "embeddings.norm.bias"], 1e-5)
    li = 0
    for s, depth in enumerate(cfg.depths):
        for b in range(depth):
            L = sw + f"encoder.layers.{s}.blocks.{b}."
            ks = None if keep_scales is None else keep_scales[li]
            x = swin_layer(p, L, x, H, W, cfg.num_heads[s], cfg.window_size, 0 if b % 2 == 0 else cfg.window_size // 2,
                           cfg.layer_norm_eps, ks)
            li += 1
        if s < len(cfg.depths) - 1:
            x = patch_merge(p, sw + f"encoder.layers.{s}.downsample.", x, H, W)
            H, W, C = H // 2, W // 2, 2 * C
    x = F.layer_norm(x, (C,), p[sw + "layernorm.weight"], p[sw + "layernorm.bias"], cfg.layer_norm_eps)
    pooled = x.mean(dim=1)
    return pooled @ p[prefix + "classifier.weight"].t() + p[prefix + "classifier.bias"]


def vit_encoder_forward(p: dict, cfg: SwinCfg, pixels: torch.Tensor, prefix: str = "cv_encoder.", keep_scales=None):
    """``Vit_Encoder.forward`` ``V/model/encoders.py:30-31``: exact (erf) GELU of the classifier output."""
    z = swin_forward(p, cfg, pixels, prefix + "image_net.", keep_scales)
    return 0.5 * z * (1.0 + torch.erf(z / math.sqrt(2.0)))
