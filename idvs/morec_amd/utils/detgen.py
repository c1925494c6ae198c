"""Deterministic, framework-independent tensor generator.

SURVEY.md §8(c) G6 asks for weights that both the golden-capture script (run once in the
build container against the imported reference) and the GPU-side tests can re-create
bit-identically without committing hundreds of MB.  The generator is a counter hash
(splitmix64 finaliser) over ``(seed(name), element index)`` -> 24-bit mantissa uniform in
[0, 1) -> affine map.  Pure numpy integer arithmetic: independent of torch / platform RNG.
"""
from __future__ import annotations

import hashlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _seed_of(name: str, seed: int) -> np.uint64:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _u01(name: str, seed: int, n: int, stream: int = 0) -> np.ndarray:
    base = _seed_of(name, seed)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(stream)) ^ base
    bits = _splitmix64(idx) >> np.uint64(40)  # top 24 bits
    return bits.astype(np.float64) * (1.0 / 16777216.0)


def det_uniform(name: str, shape, lo: float = -1.0, hi: float = 1.0, seed: int = 12345) -> np.ndarray:
    """float32 array, uniform in [lo, hi), a pure function of (name, shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = _u01(name, seed, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def det_normal(name: str, shape, std: float = 1.0, seed: int = 12345) -> np.ndarray:
    """float32 array ~ N(0, std^2) by Box-Muller over two hash streams (float64 maths, rounded once)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = _u01(name, seed, n, 0)
    u2 = _u01(name, seed, n, 1)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    z = r * np.cos(2.0 * np.pi * u2)
    return (std * z).astype(np.float32).reshape(shape)


def det_randint(name: str, shape, lo: int, hi: int, seed: int = 12345) -> np.ndarray:
    """int64 array uniform in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = _u01(name, seed, n)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64).reshape(shape)


def det_param(key: str, shape, seed: int = 12345) -> np.ndarray:
    """Deterministic parameter value for a ``state_dict`` key (used by goldens, tests, smoke):
    LayerNorm gains ~ 1 +- 0.1, every bias ~ N(0, 0.02^2), matrices/embeddings ~ N(0, s^2) with
    ``s = min(0.08, 0.6/sqrt(shape[-1]))`` (0.0217 at H=768, BERT's 0.02 class)."""
    shape = tuple(int(s) for s in shape)
    low = key.lower()
    swin_ln = low.endswith((".layernorm_before.weight", ".layernorm_after.weight", ".norm.weight"))   # Swin's LayerNorms
    if low.endswith("layernorm.weight") or low.endswith("layer_norm.weight") or swin_ln:
        return (1.0 + 0.1 * det_uniform(key, shape, seed=seed)).astype(np.float32)
    if low.endswith(".bias"):
        return det_normal(key, shape, std=0.02, seed=seed)
    std = min(0.08, 0.6 / float(np.sqrt(shape[-1])))
    return det_normal(key, shape, std=std, seed=seed)
