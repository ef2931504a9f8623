"""Deterministic synthetic frames: integer-only, identical in numpy (oracle side) and torch (GPU side).

No std::mt19937 / uniform_int_distribution: the noise is a 32-bit integer hash of (x, y, seed) whose intermediate
products stay below 2^63, so int64 arithmetic gives the same bits in numpy, torch-CPU and torch-ROCm.
"""
from __future__ import annotations

import numpy as np

_M32 = 0xFFFFFFFF


def _hash(xp, x, y, seed):
    h = (x * 0x1E3779B1 + y * 0x05EBCA77 + seed * 0x42B2AE3D + 0x165667B1) & _M32
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & _M32
    h = h ^ (h >> 12)
    h = (h * 0x297A2D39) & _M32
    h = h ^ (h >> 15)
    return h


def _frame(xp, x, y, seed, bits, kind):
    """x, y: int64 index grids; returns int64 sample values in [0, 2^bits)."""
    maxval = (1 << bits) - 1
    h = _hash(xp, x, y, seed)
    if bits <= 8:
        period, amp, noise_span = 1 << (bits + 0), 1, 3
        t = ((3 * x + 2 * y) >> 4) % period
        half = period // 2
        tri = xp.where(t < half, t, period - 1 - t) * 2 * amp
        noise = (h % (2 * noise_span + 1)) - noise_span
    else:
        # "medical style": slow ramp over the full range with wider noise
        period = 1 << (bits + 1)
        t = ((5 * x + 3 * y) * 4) % period
        half = period // 2
        tri = xp.where(t < half, t, period - 1 - t)
        span = 20 if bits >= 12 else 6
        noise = (h % (2 * span + 1)) - span
    v = tri + noise
    if kind == "mixed":
        # flat 32x32 patches (run mode), one block in eight, value taken from the block index
        bx, by = x >> 5, y >> 5
        flat = ((bx * 5 + by * 3 + seed) % 8) == 0
        v = xp.where(flat, ((bx * 37 + by * 101 + seed * 13) % (maxval + 1)), v)
    elif kind == "zero":
        v = v * 0
    elif kind == "noise":
        v = h % (maxval + 1)
    elif kind == "hard":
        span = max(2, (maxval + 1) // 8)
        v = tri + (h % (2 * span + 1)) - span
    if hasattr(xp, "clip"):
        v = xp.clip(v, 0, maxval)
    else:
        v = xp.clamp(v, 0, maxval)
    return v


def frame_numpy(width, height, seed=1, bits=8, components=1, kind="gradient", interleaved=True):
    """uint8/uint16 ndarray: (H, W) for 1 component; (H, W, C) if interleaved else (C, H, W)."""
    y, x = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    dt = np.uint8 if bits <= 8 else np.uint16
    planes = [_frame(np, x, y, seed + 7919 * c, bits, kind).astype(dt) for c in range(components)]
    if components == 1:
        return planes[0]
    return np.stack(planes, axis=2 if interleaved else 0)


def frames_torch(count, width, height, seed0, bits=8, kind="gradient", device="cuda"):
    """(count, H, W) uint8 / int16-viewed-as-uint16 tensor on `device`; frame f uses seed seed0 + f."""
    import torch

    class _XP:
        where = staticmethod(torch.where)
        clamp = staticmethod(torch.clamp)

    y = torch.arange(height, dtype=torch.int64, device=device).view(height, 1).expand(height, width)
    x = torch.arange(width, dtype=torch.int64, device=device).view(1, width).expand(height, width)
    dt = torch.uint8 if bits <= 8 else torch.int16
    out = torch.empty((count, height, width), dtype=dt, device=device)
    for f in range(count):
        v = _frame(_XP, x, y, seed0 + f, bits, kind)
        if bits <= 8:
            out[f] = v.to(torch.uint8)
        else:
            out[f] = v.to(torch.int32).to(torch.int16)  # same 16 bits; torch has no native uint16 arithmetic
    return out
