"""Counter-based RNG used for every synthetic tensor (weights, frames).

Design goals: (1) element i of tensor `name` under `seed` is a pure function of
(seed, name, i) -- no stream state, so any rank / any process regenerates the
same bytes; (2) bit-exact across platforms: only integer arithmetic plus exact
float64 scalings are used (no libm log/cos), so goldens generated in one
container match tensors regenerated on the GPU box.

normal():  Irwin-Hall sum of twelve 16-bit uniforms (3 x splitmix64 words),
           mean 0, variance (65536^2-1)/65536^2 ~= 1.
uniform(): 53-bit mantissa uniform in [0,1).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser (vectorised, wrap-around uint64 arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def key_for(seed: int, name: str) -> np.uint64:
    """64-bit stream key from (seed, tensor name): FNV-1a over the name, mixed with the seed."""
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    k = (h ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    return _mix(np.array([k], dtype=np.uint64))[0]


def _words(key: np.uint64, n: int, lane: int, nlanes: int) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        ctr = key + (idx * np.uint64(nlanes) + np.uint64(lane) + np.uint64(1)) * _GOLDEN
    return _mix(ctr)


def uniform(seed: int, name: str, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    w = _words(key_for(seed, name), n, 0, 1)
    u = (w >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).astype(np.float64)


def normal(seed: int, name: str, n: int) -> np.ndarray:
    """~N(0,1) float64 values that are exact multiples of 2^-16."""
    key = key_for(seed, name)
    s = np.zeros(n, dtype=np.int64)
    for lane in range(3):
        w = _words(key, n, lane, 3)
        for sh in (0, 16, 32, 48):
            s += ((w >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    return (s - 6 * 65535).astype(np.float64) / 65536.0


def frame(seed: int, h: int, w: int, c: int = 3) -> np.ndarray:
    """Synthetic normalised camera frame, NCHW fp32 ~N(0,1) (reference smoke input: model.py:395)."""
    return normal(seed, "frame", c * h * w).astype(np.float32).reshape(1, c, h, w)
