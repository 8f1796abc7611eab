"""Portable, torch-free seeded generator (splitmix64 -> uniform -> Box-Muller).

Used to build weights, synthetic tiles and dropout masks that are bit-identical in
every process that needs them (golden-fixture generator in the build container,
parity tests and bench.py on the GPU box), without relying on torch's RNG streams
and without shipping 163 MB of weights.  SURVEY.md section 7 step 1 / section 8(d).
"""
from __future__ import annotations

import zlib

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """Counter-based splitmix64: value i of stream `seed` (vectorised, wrap-around uint64)."""
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(1)) * _GOLDEN + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def name_seed(name: str, seed: int) -> int:
    """Stable per-tensor stream id derived from a tensor name (crc32) and a base seed."""
    return ((zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF) << 20) ^ (seed * 0x1000193 + 0x5BD1E995)


def uniform(n: int, seed: int) -> np.ndarray:
    """n float64 values in the open interval (0, 1)."""
    z = _splitmix64(np.arange(n, dtype=np.uint64), seed)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(n: int, seed: int) -> np.ndarray:
    """n float64 standard-normal values (Box-Muller on two decorrelated streams)."""
    u1 = uniform(n, seed)
    u2 = uniform(n, seed ^ 0x5DEECE66D1234567)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def normal_f32(shape, seed: int, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape))
    return (normal(n, seed) * std + mean).astype(np.float32).reshape(shape)


def uniform_f32(shape, seed: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    return (uniform(n, seed) * (hi - lo) + lo).astype(np.float32).reshape(shape)


def bernoulli_keep(shape, seed: int, p_drop: float) -> np.ndarray:
    """Keep-mask (1.0 = kept) with drop probability p_drop, float32."""
    n = int(np.prod(shape))
    return (uniform(n, seed) >= p_drop).astype(np.float32).reshape(shape)
