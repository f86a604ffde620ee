"""Stand-in for the two NATTEN (<=0.17) functional ops the reference calls -- TEST INFRASTRUCTURE.

Used ONLY inside the build container by ``oracle/make_golden.py`` so that the reference's own Python
(/root/reference/src/...) can be imported and executed on CPU: the reference hard-imports
``natten.functional.na2d_qk / na2d_av`` (src/layers/attentions.py:6-11,20,24) and NATTEN (SHI-Labs,
pinned natten==0.17.4+torch240cu118 in docs/INSTALL.md:7) is not vendored, not installed and not
installable here.  The semantics below restate NATTEN's published naive CPU kernels
(``get_window_start`` + pointwise neighbourhood, row-major (ki, kj) key order).  Because the real
NATTEN is absent, results at this boundary are "parity unpinned"; everything else the reference runs
(conv stem, GroupNorm, pooling, RoPE, resize, scale, softmax, head split) is its own code.

This file is never imported by the product (naf_amd/) and never runs on the GPU box.
"""
from __future__ import annotations

import sys
import types

import numpy as np
import torch


def _window_starts(L: int, k: int, dil: int) -> np.ndarray:
    """Vectorised NATTEN<=0.17 get_window_start for all i in [0, L)."""
    if k * dil > L:
        raise ValueError(f"Input axis {L} must be >= kernel_size * dilation = {k * dil}")
    r = k // 2
    i = np.arange(L, dtype=np.int64)
    if dil <= 1:
        return np.maximum(i - r, 0) + (i + r >= L) * (L - i - r - 1)
    ni = i - r * dil
    m = i % dil
    a = (L // dil) * dil
    b = L - a
    right = np.where(m < b, L - b + m - 2 * r * dil, a + m - k * dil)
    out = np.where(ni < 0, m, np.where(i + r * dil >= L, right, ni))
    return out


def _neigh(L: int, k: int, dil: int) -> torch.Tensor:
    s = _window_starts(L, k, dil)
    return torch.from_numpy(s[:, None] + dil * np.arange(k, dtype=np.int64)[None, :])   # [L, k]


def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def na2d_qk(query, key, kernel_size, dilation=1, **_):
    """query/key [B, heads, H, W, D] -> [B, heads, H, W, ky*kx]."""
    ky, kx = _pair(kernel_size)
    dy, dx = _pair(dilation)
    B, n, H, W, D = query.shape
    iy, ix = _neigh(H, ky, dy), _neigh(W, kx, dx)
    out = query.new_empty(B, n, H, W, ky * kx)
    step = max(1, 16384 // max(W, 1) // max(ky * kx, 1) * 4)
    for r0 in range(0, H, step):
        r1 = min(H, r0 + step)
        kg = key[:, :, iy[r0:r1]][:, :, :, :, ix]                      # b n R ky W kx d
        kg = kg.permute(0, 1, 2, 4, 3, 5, 6).reshape(B, n, r1 - r0, W, ky * kx, D)
        out[:, :, r0:r1] = torch.einsum("bnrwd,bnrwkd->bnrwk", query[:, :, r0:r1], kg)
    return out


def na2d_av(attn, value, kernel_size, dilation=1, **_):
    """attn [B, heads, H, W, ky*kx], value [B, heads, H, W, D] -> [B, heads, H, W, D]."""
    ky, kx = _pair(kernel_size)
    dy, dx = _pair(dilation)
    B, n, H, W, D = value.shape
    iy, ix = _neigh(H, ky, dy), _neigh(W, kx, dx)
    out = value.new_empty(B, n, H, W, D)
    step = max(1, 16384 // max(W, 1) // max(ky * kx, 1) * 4)
    for r0 in range(0, H, step):
        r1 = min(H, r0 + step)
        vg = value[:, :, iy[r0:r1]][:, :, :, :, ix]
        vg = vg.permute(0, 1, 2, 4, 3, 5, 6).reshape(B, n, r1 - r0, W, ky * kx, D)
        out[:, :, r0:r1] = torch.einsum("bnrwk,bnrwkd->bnrwd", attn[:, :, r0:r1], vg)
    return out


def install():
    """Register ``natten`` / ``natten.functional`` in sys.modules (container-only helper)."""
    if "natten" in sys.modules and getattr(sys.modules["natten"], "__naf_shim__", False):
        return
    pkg = types.ModuleType("natten")
    fn = types.ModuleType("natten.functional")
    fn.na2d_qk, fn.na2d_av = na2d_qk, na2d_av
    pkg.functional = fn
    pkg.__naf_shim__ = True
    pkg.__version__ = "0.17.4+naf-oracle-shim"
    sys.modules["natten"] = pkg
    sys.modules["natten.functional"] = fn
