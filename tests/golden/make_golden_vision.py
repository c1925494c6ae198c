#!/usr/bin/env python
"""Capture golden vectors for the VISION hot path from the imported reference.

Run once in the build container:   python tests/golden/make_golden_vision.py

Imports ``/root/reference/inbatch_sasrec_e2e_vision``'s ``model`` package unmodified (CPU, fp32) together with the
installed HuggingFace ``SwinForImageClassification`` (the third-party class the reference builds at ``V/run.py:47-54``),
feeds deterministic synthetic pixels / weights (``det_param``: only inputs that cannot be regenerated and numeric
OUTPUTS are stored) and writes ``g11_swin_micro.npz`` + ``g12_vision_keys.json`` next to this file.
A separate script from ``make_golden.py`` because both reference variants name their package ``model``.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/inbatch_sasrec_e2e_vision")

from idvs.morec_amd.utils.detgen import det_normal, det_param, det_randint  # noqa: E402

from model import Model as RefModel  # noqa: E402  (reference package)
from model.encoders import Vit_Encoder as RefVitEncoder  # noqa: E402
from transformers import SwinConfig, SwinForImageClassification  # noqa: E402

torch.set_num_threads(8)

MICRO = dict(image_size=56, patch_size=4, num_channels=3, embed_dim=32, depths=[2, 2], num_heads=[1, 2], window_size=7,
             mlp_ratio=4.0, drop_path_rate=0.1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
# three stages, resolution 28 -> 14 -> 7: two shifted stages (4 and 1... windows per side) + the window == resolution case
MICRO3 = dict(image_size=112, patch_size=4, num_channels=3, embed_dim=32, depths=[2, 2, 2], num_heads=[1, 2, 4],
              window_size=7, mlp_ratio=4.0, drop_path_rate=0.1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def load_det(module, seed=12345):
    sd = module.state_dict()
    new = {k: torch.from_numpy(det_param(k, tuple(v.shape), seed=seed)) for k, v in sd.items() if v.dtype.is_floating_point}
    module.load_state_dict(new, strict=False)
    return module


def build_swin(kw, D):
    cfg = SwinConfig(attn_implementation="eager", **kw)
    net = SwinForImageClassification(cfg)
    net.classifier = torch.nn.Linear(net.classifier.in_features if hasattr(net.classifier, "in_features") else net.swin.num_features, D)  # V/run.py:50-51
    return net


def synth_batch(name, B, S, item_num):
    ids = np.zeros((B, S + 1), dtype=np.int64)
    log_mask = np.zeros((B, S), dtype=np.float32)
    for b in range(B):
        L = int(det_randint(f"{name}.len{b}", (1,), 3, S + 2)[0])
        seq = det_randint(f"{name}.seq{b}", (L,), 1, item_num + 1)
        if L >= 4:
            seq[-2] = seq[0]
            if b > 0:
                seq[1] = ids[b - 1, -1]
        ids[b, S + 1 - L:] = seq
        log_mask[b, S + 1 - L:] = 1.0
        log_mask[b] = (ids[b, :-1] != 0).astype(np.float32)
    return ids, log_mask


def g11(out):
    res = {}
    for tag, kw, N, D in [("m2", MICRO, 3, 48), ("m3", MICRO3, 2, 64)]:
        net = build_swin(kw, D)
        enc = RefVitEncoder(net)
        wrap = torch.nn.Module()
        wrap.cv_encoder = enc                       # keys get the reference's ``cv_encoder.image_net.`` prefix
        load_det(wrap)
        wrap.eval()
        R0 = kw["image_size"]
        x = torch.from_numpy(det_normal(f"g11{tag}.x", (N, 3, R0, R0), std=1.0))
        Rm = torch.from_numpy(det_normal(f"g11{tag}.R", (N, D), std=1.0))
        logits = net(x)[0]
        y = enc(x)
        wrap.zero_grad()
        (enc(x) * Rm).sum().backward()
        res[f"{tag}.cfg"] = np.array([R0, kw["patch_size"], kw["embed_dim"], kw["window_size"], N, D])
        res[f"{tag}.depths"], res[f"{tag}.heads"] = np.array(kw["depths"]), np.array(kw["num_heads"])
        res[f"{tag}.logits"], res[f"{tag}.y"] = logits.detach().numpy(), y.detach().numpy()
        named = dict(wrap.named_parameters())
        for k, p in named.items():
            res[f"{tag}.grad_norm.{k}"] = np.float64(p.grad.double().norm().item())
        pre = "cv_encoder.image_net."
        keep = [pre + "classifier.weight", pre + "swin.layernorm.weight",
                pre + "swin.embeddings.patch_embeddings.projection.weight", pre + "swin.embeddings.norm.bias",
                pre + "swin.encoder.layers.0.blocks.1.attention.relative_position_bias.relative_position_bias_table",
                pre + "swin.encoder.layers.0.blocks.1.attention.q_proj.weight",
                pre + "swin.encoder.layers.0.blocks.1.attention.k_proj.bias",
                pre + "swin.encoder.layers.0.blocks.0.mlp.fc1.weight",
                pre + "swin.encoder.layers.0.downsample.reduction.weight",
                pre + "swin.encoder.layers.0.downsample.norm.weight",
                pre + "swin.encoder.layers.1.blocks.0.attention.relative_position_bias.relative_position_bias_table",
                pre + "swin.encoder.layers.1.blocks.1.attention.o_proj.weight",
                pre + "swin.encoder.layers.1.blocks.1.mlp.fc2.bias"]
        for k in keep:
            res[f"{tag}.grad.{k}"] = named[k].grad.numpy().copy()
        print("g11", tag, "y norm", float(y.norm()))

    # full vision Model.forward (V/model/model.py:35-73) with the micro Swin tower
    S, D, item_num, B = 5, 48, 40, 3
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 CV_model_load="swin_micro")
    pop = np.abs(det_normal("pop.g11", (item_num + 1,), std=1.0)) + 0.05
    pop = (pop / pop.sum()).astype(np.float32)
    m = RefModel(args, item_num, True, build_swin(MICRO, D), pop.tolist())
    load_det(m)
    m.eval()
    ids, log_mask = synth_batch("g11f", B, S, item_num)
    images = det_normal("g11f.images", (item_num + 1, 3, 56, 56), std=1.0).astype(np.float32)
    images[0] = 0.0
    px = torch.from_numpy(images[ids.reshape(-1)])
    m.zero_grad()
    loss = m(torch.from_numpy(ids).view(-1), px, torch.from_numpy(log_mask), "cpu")
    loss.backward()
    res["full.cfg"] = np.array([S, D, item_num, B])
    res["full.ids"], res["full.log_mask"], res["full.pop"] = ids, log_mask, pop
    res["full.loss"] = np.float32(loss.item())
    for k, p in m.named_parameters():
        if p.grad is not None:
            res[f"full.grad_norm.{k}"] = np.float64(p.grad.double().norm().item())
    print("g11 full loss", loss.item())
    np.savez_compressed(os.path.join(out, "g11_swin_micro.npz"), **res)

    # state_dict / named_parameters surface with the real Swin-T config (V/run.py:58-60 freezes by parameter INDEX)
    cfg_t = SwinConfig.from_pretrained("/root/reference/pretrained_models/swin_tiny").to_dict()
    kw_t = {k: cfg_t[k] for k in ["image_size", "patch_size", "num_channels", "embed_dim", "depths", "num_heads", "window_size",
                                  "mlp_ratio", "drop_path_rate", "hidden_dropout_prob", "attention_probs_dropout_prob",
                                  "layer_norm_eps"]}
    kw_t["depths"] = [2, 2, 2, 2]     # per-block key pattern is what matters; keeps the fixture small
    args = types.SimpleNamespace(max_seq_len=10, embedding_dim=2048, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                                 CV_model_load="swin_tiny")
    m = RefModel(args, 100, True, build_swin(kw_t, 2048), [1.0] * 101)
    keys = dict(state_dict=[[k, list(v.shape)] for k, v in m.state_dict().items()],
                named_parameters=[k for k, _ in m.named_parameters()], swin_tiny_config=kw_t)
    with open(os.path.join(out, "g12_vision_keys.json"), "w") as f:
        json.dump(keys, f)
    print("g12 keys", len(keys["state_dict"]))


def g13(out, name="swin_tiny", fname="g13_swin_tiny_scalars.npz", tag="g13", B=2):
    """Full-size Swin-T (config from pretrained_models/swin_tiny) inside the reference vision ``Model``: scalars only (loss, probe
    elements of the item vectors, gradient norms); weights come from ``det_param`` on both sides.
    ``name='swin_base'`` (g15): the same capture with pretrained_models/swin_base/config.json -- BASELINE.json configs[4]
    (embed 128, depths 2/2/18/2, heads 4/8/16/32), launcher V/run.py:47-54."""
    res = {}
    cfg_t = SwinConfig.from_pretrained(f"/root/reference/pretrained_models/{name}").to_dict()
    kw = {k: cfg_t[k] for k in ["image_size", "patch_size", "num_channels", "embed_dim", "depths", "num_heads", "window_size",
                                "mlp_ratio", "drop_path_rate", "layer_norm_eps"]}
    kw.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    S, D, item_num = 3, 256, 12
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 CV_model_load=name)
    pop = np.abs(det_normal(f"pop.{tag}", (item_num + 1,), std=1.0)) + 0.05
    pop = (pop / pop.sum()).astype(np.float32)
    m = RefModel(args, item_num, True, build_swin(kw, D), pop.tolist())
    load_det(m)
    m.eval()
    ids, log_mask = synth_batch(tag, B, S, item_num)
    images = det_normal(f"{tag}.images", (item_num + 1, 3, 224, 224), std=1.0).astype(np.float32)
    images[0] = 0.0
    px = torch.from_numpy(images[ids.reshape(-1)])
    m.zero_grad()
    loss = m(torch.from_numpy(ids).view(-1), px, torch.from_numpy(log_mask), "cpu")
    loss.backward()
    with torch.no_grad():
        vec = m.cv_encoder(px)
    res["cfg"] = np.array([S, D, item_num, B])
    res["ids"], res["log_mask"], res["pop"] = ids, log_mask, pop
    res["loss"] = np.float32(loss.item())
    res["item_vec_probe"] = vec[:, :8].numpy()
    for k, p in m.named_parameters():
        if p.grad is not None:
            res[f"grad_norm.{k}"] = np.float64(p.grad.double().norm().item())
    np.savez_compressed(os.path.join(out, fname), **res)
    print(tag, name, "loss", loss.item())


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.only in ("", "g11"):
        g11(HERE)
    if a.only in ("", "g13"):
        g13(HERE)
    if a.only in ("", "g15"):
        g13(HERE, name="swin_base", fname="g15_swin_base_scalars.npz", tag="g15", B=2)
