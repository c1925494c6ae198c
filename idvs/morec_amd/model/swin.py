"""``HipSwinForImageClassification``: parameter container with HuggingFace ``SwinForImageClassification``'s module tree
and ``state_dict`` keys (the reference builds it at ``V/run.py:47-54`` -- ``from_pretrained`` then a fresh
``classifier = Linear(num_features, embedding_dim)`` -- and reaches it as ``model.module.cv_encoder.image_net``,
``V/run.py:138``).  Forward arithmetic is in ``libmorec_hip.so`` (``functional.SwinEncoderFn``); this class only owns
the weights so that a HF checkpoint loads key for key (``from_hf``), including the reference's pinned
``transformers==4.20.1`` key names (``remap_legacy_swin_keys``)."""
import re
import types

import torch
import torch.nn as nn

from ..swin_engine import SwinShape


class _RelBias(nn.Module):
    def __init__(self, ws, heads):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))


class _Attention(nn.Module):
    def __init__(self, C, heads, ws):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = nn.Linear(C, C), nn.Linear(C, C), nn.Linear(C, C), nn.Linear(C, C)
        self.relative_position_bias = _RelBias(ws, heads)


class _MLP(nn.Module):
    def __init__(self, C, I):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(C, I), nn.Linear(I, C)


class _Layer(nn.Module):
    def __init__(self, C, heads, ws, I, eps):
        super().__init__()
        self.attention = _Attention(C, heads, ws)
        self.layernorm_before = nn.LayerNorm(C, eps=eps)
        self.layernorm_after = nn.LayerNorm(C, eps=eps)
        self.mlp = _MLP(C, I)


class _Merging(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.reduction = nn.Linear(4 * C, 2 * C, bias=False)
        self.norm = nn.LayerNorm(4 * C)


class _Stage(nn.Module):
    def __init__(self, C, depth, heads, ws, I, eps, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([_Layer(C, heads, ws, I, eps) for _ in range(depth)])
        if downsample:
            self.downsample = _Merging(C)


class _Encoder(nn.Module):
    def __init__(self, s: SwinShape):
        super().__init__()
        n = len(s.depths)
        self.layers = nn.ModuleList([_Stage(s.embed_dim * 2 ** i, s.depths[i], s.num_heads[i], s.window_size,
                                            int(s.mlp_ratio * s.embed_dim * 2 ** i), s.layer_norm_eps, i < n - 1) for i in range(n)])


class _PatchEmbeddings(nn.Module):
    def __init__(self, s: SwinShape):
        super().__init__()
        self.projection = nn.Conv2d(s.num_channels, s.embed_dim, kernel_size=s.patch_size, stride=s.patch_size)


class _Embeddings(nn.Module):
    def __init__(self, s: SwinShape):
        super().__init__()
        self.patch_embeddings = _PatchEmbeddings(s)
        self.norm = nn.LayerNorm(s.embed_dim)


class _SwinModel(nn.Module):
    def __init__(self, s: SwinShape):
        super().__init__()
        self.embeddings = _Embeddings(s)
        self.encoder = _Encoder(s)
        self.num_features = s.embed_dim * 2 ** (len(s.depths) - 1)
        self.layernorm = nn.LayerNorm(self.num_features, eps=s.layer_norm_eps)


_LEGACY = [   # transformers 4.20.1 (README.md:44) -> installed names
    (r"\.attention\.self\.query\.", ".attention.q_proj."), (r"\.attention\.self\.key\.", ".attention.k_proj."),
    (r"\.attention\.self\.value\.", ".attention.v_proj."), (r"\.attention\.output\.dense\.", ".attention.o_proj."),
    (r"\.attention\.self\.relative_position_bias_table", ".attention.relative_position_bias.relative_position_bias_table"),
    (r"\.intermediate\.dense\.", ".mlp.fc1."), (r"(blocks\.\d+)\.output\.dense\.", r"\1.mlp.fc2."),
]


def remap_legacy_swin_keys(sd: dict) -> dict:
    """Rename a ``transformers==4.20.1``-era Swin ``state_dict`` (what the reference's checkpoints hold) to the installed
    layout; ``relative_position_index`` buffers are dropped (recomputed from the window size)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("relative_position_index"):
            continue
        for pat, rep in _LEGACY:
            k = re.sub(pat, rep, k)
        out[k] = v
    return out


class HipSwinForImageClassification(nn.Module):
    def __init__(self, shape: SwinShape, num_labels: int = 1000, initializer_range: float = 0.02):
        super().__init__()
        self.shape = shape
        self.config = types.SimpleNamespace(image_size=shape.image_size, patch_size=shape.patch_size, embed_dim=shape.embed_dim,
                                            depths=list(shape.depths), num_heads=list(shape.num_heads),
                                            window_size=shape.window_size, mlp_ratio=shape.mlp_ratio,
                                            layer_norm_eps=shape.layer_norm_eps, drop_path_rate=shape.drop_path_rate,
                                            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, num_labels=num_labels)
        self.swin = _SwinModel(shape)
        self.classifier = nn.Linear(self.swin.num_features, num_labels)   # V/run.py:50-51 replaces it by Linear(., embedding_dim)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.trunc_normal_(m.weight, std=initializer_range)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    @staticmethod
    def from_hf(hf_model) -> "HipSwinForImageClassification":
        """Adopt a HuggingFace ``SwinForImageClassification`` (after the reference replaced its classifier): same keys
        and parameter order, weights copied."""
        c = hf_model.config
        if float(getattr(c, "hidden_dropout_prob", 0.0)) or float(getattr(c, "attention_probs_dropout_prob", 0.0)):
            raise NotImplementedError("Swin hidden / attention dropout (0.0 in every Swin config the reference ships)")
        if getattr(c, "use_absolute_embeddings", False):
            raise NotImplementedError("absolute position embeddings (off in every Swin config the reference ships)")
        shape = SwinShape(image_size=c.image_size, patch_size=c.patch_size, num_channels=c.num_channels, embed_dim=c.embed_dim,
                          depths=tuple(c.depths), num_heads=tuple(c.num_heads), window_size=c.window_size,
                          mlp_ratio=c.mlp_ratio, layer_norm_eps=c.layer_norm_eps, drop_path_rate=c.drop_path_rate)
        m = HipSwinForImageClassification(shape, num_labels=hf_model.classifier.out_features)
        sd = remap_legacy_swin_keys(hf_model.state_dict())
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not missing, missing
        for (n1, p1), (n2, p2) in zip(m.named_parameters(), hf_model.named_parameters()):
            assert n1 == n2, (n1, n2)       # same ORDER too: V/run.py:58-60 freezes parameters by index
            p1.requires_grad = p2.requires_grad
        return m
