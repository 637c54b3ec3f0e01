"""Synthetic correction vectors for benchmarks and full-size tests.

f_t(i) = uniform(-1, 1) from a counter-based generator keyed on
(seed, call index t, GLOBAL element index i) -- SURVEY.md 8(d) -- so every rank
count sees the same global vector and nothing crosses PCIe.  SplitMix64
finaliser over the 64-bit counter; top 53 bits -> [0,1).  The torch (device)
and numpy (host) versions produce identical bits.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1
_G = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def _s64(x: int) -> int:
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _key(seed: int, t: int, n_global: int) -> int:
    return ((seed + 1) * _G + t * n_global) & _M64


def fill_numpy(seed: int, t: int, lo: int, hi: int, n_global: int) -> np.ndarray:
    i = np.arange(lo, hi, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = i + np.uint64(_key(seed, t, n_global))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def fill_torch(out, seed: int, t: int, lo: int, n_global: int, chunk: int = 1 << 25):
    """Fill the 1-D float64 CUDA tensor `out` with elements lo .. lo+len(out)."""
    import torch
    n = out.numel()
    key = _s64(_key(seed, t, n_global))
    for c0 in range(0, n, chunk):
        c1 = min(c0 + chunk, n)
        z = torch.arange(lo + c0, lo + c1, dtype=torch.int64, device=out.device) + key
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _s64(_C1)
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _s64(_C2)
        z = z ^ ((z >> 31) & ((1 << 33) - 1))
        out[c0:c1] = ((z >> 11) & ((1 << 53) - 1)).to(torch.float64) * (2.0 / (1 << 53)) - 1.0
    return out
