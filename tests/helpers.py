"""Shared helpers for tests: deterministic parameters keyed by the reference's state_dict names."""
import numpy as np
import torch

from idvs.morec_amd.utils.detgen import det_param


def det_state(shapes, as_torch=True, seed=12345):
    out = {}
    for k, shp in shapes.items():
        v = det_param(k, shp, seed=seed)
        out[k] = torch.from_numpy(v) if as_torch else v
    return out


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


import os as _os

GOLDEN_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")
