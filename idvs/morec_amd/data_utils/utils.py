"""Checkpoint helpers in the reference's format (``T/data_utils/utils.py:107-114``, ``T/run.py:130-139``): ``epoch-N.pt`` is a
dict with ``model_state_dict`` (UNWRAPPED module: the keys of ``idvs.morec_amd.model.Model.state_dict()`` equal the
reference's), ``optimizer``, ``rng_state``, ``cuda_rng_state`` and ``scaler_state``."""
from __future__ import annotations

import os
import re

import torch


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def save_model(now_epoch, model, model_dir, optimizer, rng_state, cuda_rng_state, scaler=None, Log_file=None, extra=None):
    """``extra``: additional top-level entries (ignored by the reference's loader, ``T/run.py:130-139``) -- e.g. the call counter of the
    library's counter-based dropout streams, which torch's RNG state does not cover."""
    os.makedirs(model_dir, exist_ok=True)
    ckpt_path = os.path.join(model_dir, f"epoch-{now_epoch}.pt")
    torch.save({**(extra or {}), "model_state_dict": _unwrap(model).state_dict(),
                # torch.optim.AdamW or train_step.TrainStep (same state_dict form: TrainStep.optimizer_state_dict)
                "optimizer": (optimizer.optimizer_state_dict() if hasattr(optimizer, "optimizer_state_dict") else optimizer.state_dict())
                if optimizer is not None else None,
                "rng_state": rng_state, "cuda_rng_state": cuda_rng_state,
                # torch.cuda.amp.GradScaler (state_dict) or train_step.TrainStep (scaler_state_dict: its device-side loss-scale block)
                "scaler_state": ({} if scaler is None else scaler.scaler_state_dict() if hasattr(scaler, "scaler_state_dict")
                                 else scaler.state_dict())}, ckpt_path)
    if Log_file is not None:
        Log_file.info(f"Model saved to {ckpt_path}")
    return ckpt_path


def get_checkpoint(directory, ckpt_name):
    path = os.path.join(directory, ckpt_name)
    return path if os.path.exists(path) else None


def load_model(model, ckpt_path, optimizer=None, strict=True):
    """Load a reference-format checkpoint.  ``embeddings.position_ids`` (saved by transformers 4.20.1, a non-persistent
    buffer since) is dropped if present.  Returns the epoch parsed from the file name (``T/run.py:137``)."""
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    sd = {k: v for k, v in ckpt["model_state_dict"].items() if not k.endswith("embeddings.position_ids")}
    _unwrap(model).load_state_dict(sd, strict=strict)
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    m = re.split(r"[._-]", os.path.basename(ckpt_path))
    return int(m[1]) if len(m) > 1 and m[1].isdigit() else 0
