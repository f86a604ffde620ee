"""CPU oracle for the NAF forward hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  Nothing under ``naf_amd/`` imports it; the product path fails loudly without its HIP library.

What it is
----------
A plain fp32/fp64 PyTorch restatement of the reference's forward
``naf(image, lr_features, target_size)`` written from the reference's formulas (file:line cited on
each function, paths relative to /root/reference).  Every function is a straight-line tensor
program with no module state, evaluated on the CPU: the golden fixtures, bench.py's cpu_baseline
and every parity test at sizes the host finishes in seconds.  The functions create their index /
coordinate tensors on their inputs' device, so the SAME code can also be evaluated by ATen's fp32 /
fp64 device kernels: the GPU suite does that for its largest cases only (the 2048^2 stem, the G3
shard, fp64 autograd of wide heads with 11x11 ... 15x15 windows -- minutes of host time otherwise)
after ``tests/test_gpu_parity.py::test_oracle_on_the_device_equals_the_oracle_on_the_host`` has held
the device evaluation to the host evaluation on that very box.  Nothing hand-written runs in it.

Parity status
-------------
* conv stem, GroupNorm, SiLU, pooling, RoPE, head split, nearest-exact mapping, scale, softmax,
  wiring: PINNED -- checked in this container against the imported reference's own code
  (``oracle/make_golden.py``; fixtures committed under ``tests/golden``).
* the neighbourhood gather itself (which keys a query attends to) lives in NATTEN
  (SHI-Labs, pins natten==0.17.4+torch240cu118 / 0.17.3 / 0.20.1, docs/INSTALL.md:7,16,20), which is
  neither vendored in the reference nor installable here, and the reference has no golden vectors
  for it:  **parity unpinned at the NATTEN boundary**.  ``natten_window_start`` restates NATTEN
  <=0.17's published ``get_window_start``; it is pinned only by NATTEN-independent identities
  (tests/test_oracle.py: full-window == dense SDPA, constant-V, unfold interior, two independent
  derivations -- dilated hi-res form vs low-res window form -- and, since round 6, the definition
  of dilation itself: delta independent undilated neighbourhood attentions on the strided sub-grids,
  for every axis length / window / dilation NATTEN accepts, remainder branch included).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# deterministic, platform-independent input generator (integer hash -> Irwin-Hall "normal")
# --------------------------------------------------------------------------------------------
def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_normal(shape: Sequence[int], seed: int, scale: float = 1.0) -> Tensor:
    """Unit-variance pseudo-normal tensor that is bit-identical on every machine.

    Sum of four 16-bit uniforms taken from one splitmix64 word (Irwin-Hall, exact in float64, no
    transcendental functions), centred and scaled to variance 1.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        w = _splitmix64(idx)
    m = np.uint64(0xFFFF)
    s = ((w & m) + ((w >> np.uint64(16)) & m) + ((w >> np.uint64(32)) & m) + ((w >> np.uint64(48)) & m))
    x = (s.astype(np.float64) - 2.0 * 65535.0) / (65536.0 * math.sqrt(4.0 / 12.0))
    return torch.from_numpy((x * scale).astype(np.float32).reshape(tuple(shape)))


# --------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------
def rope_periods(dim: int, heads_rope: int, base: float = 100.0) -> Tensor:
    """src/layers/rope.py:128-135: periods[t] = base ** (2 t / (D_head/2)), t < D_head/4."""
    d_head = dim // heads_rope
    return base ** (2 * torch.arange(d_head // 4, dtype=torch.float32) / (d_head // 2))


def make_params(dim: int = 256, heads_rope: int = 4, img_layers: int = 2, rope_base: float = 100.0,
                seed: int = 0, in_channels: int = 3) -> Dict[str, Tensor]:
    """A full NAF ``state_dict`` (same keys / shapes as src/model/naf.py:72-102 creates) filled by
    ``hash_normal`` with fan-in scaling; GroupNorm affine is perturbed away from (1, 0) so that the
    affine path is exercised."""
    hid = dim // 2
    p: Dict[str, Tensor] = {}
    s = seed * 1000
    for branch, ks in (("encoder", 1), ("sem_encoder", 3)):
        pre = f"image_encoder.{branch}"
        s += 1
        p[f"{pre}.0.weight"] = hash_normal((hid, in_channels, ks, ks), s, 1.0 / math.sqrt(in_channels * ks * ks))
        s += 1
        p[f"{pre}.0.bias"] = hash_normal((hid,), s, 0.1)
        for blk in range(1, img_layers + 1):
            for j in (1, 2):
                s += 1
                p[f"{pre}.{blk}.norm{j}.weight"] = 1.0 + hash_normal((hid,), s, 0.1)
                s += 1
                p[f"{pre}.{blk}.norm{j}.bias"] = hash_normal((hid,), s, 0.1)
                s += 1
                p[f"{pre}.{blk}.conv{j}.weight"] = hash_normal((hid, hid, ks, ks), s, 1.0 / math.sqrt(hid * ks * ks))
                s += 1
                p[f"{pre}.{blk}.conv{j}.bias"] = hash_normal((hid,), s, 0.1)
    p["image_encoder.rope.periods"] = rope_periods(dim, heads_rope, rope_base)
    return p


# --------------------------------------------------------------------------------------------
# a3: conv stem   (src/layers/convolutions.py:6-92, src/model/naf.py:26-33)
# --------------------------------------------------------------------------------------------
def _conv_reflect(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """nn.Conv2d(padding=ks//2, padding_mode='reflect') -- convolutions.py:68-75, 23-30."""
    pad = w.shape[-1] // 2
    if pad:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    return F.conv2d(x, w, b)


def enc_block(x: Tensor, p: Dict[str, Tensor], pre: str, num_groups: int = 8) -> Tensor:
    """EncBlock.forward with residual=False (convolutions.py:52-64): GN -> SiLU -> conv -> GN -> SiLU
    -> conv; the shortcut is never added (convolutions.py:62-64, residual flag False at :90)."""
    x = F.group_norm(x, num_groups, p[f"{pre}.norm1.weight"], p[f"{pre}.norm1.bias"], eps=1e-5)
    x = F.silu(x)
    x = _conv_reflect(x, p[f"{pre}.conv1.weight"], p[f"{pre}.conv1.bias"])
    x = F.group_norm(x, num_groups, p[f"{pre}.norm2.weight"], p[f"{pre}.norm2.bias"], eps=1e-5)
    x = F.silu(x)
    x = _conv_reflect(x, p[f"{pre}.conv2.weight"], p[f"{pre}.conv2.bias"])
    return x


def conv_branch(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """encoder(...) = nn.Sequential(Conv2d, EncBlock * num_layers) -- convolutions.py:67-92."""
    x = _conv_reflect(x, p[f"{pre}.0.weight"], p[f"{pre}.0.bias"])
    blk = 1
    while f"{pre}.{blk}.conv1.weight" in p:
        x = enc_block(x, p, f"{pre}.{blk}")
        blk += 1
    return x


def conv_stem(image: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """cat[1x1 branch, 3x3-reflect branch] on channels -- naf.py:31-33."""
    return torch.cat([conv_branch(image, p, "image_encoder.encoder"),
                      conv_branch(image, p, "image_encoder.sem_encoder")], dim=1)


# --------------------------------------------------------------------------------------------
# a4: RoPE   (src/layers/rope.py:15-34, 84-105, 137-174), eval-mode coordinates only
# --------------------------------------------------------------------------------------------
def rope_angles(H: int, W: int, periods: Tensor) -> Tensor:
    """[H*W, D_head] angles.  rope.py:98-105 (normalize 'separate'), :139-143."""
    dt = periods.dtype
    ch = torch.arange(0.5, H, dtype=dt, device=periods.device) / H
    cw = torch.arange(0.5, W, dtype=dt, device=periods.device) / W
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    ang = 2 * math.pi * coords[:, :, None] / periods[None, None, :]
    return ang.flatten(1, 2).tile(2)


def rope(x: Tensor, periods: Tensor, heads: int) -> Tensor:
    """x [B, heads*D, H, W] -> same.  out = x*cos + rotate_half(x)*sin, rotate_half pairs channel t
    with t + D/2 inside a head (rope.py:15-19,34); every head uses the same angles (:155-157)."""
    B, C, H, W = x.shape
    D = C // heads
    ang = rope_angles(H, W, periods.to(torch.float32) if x.dtype != torch.float64 else periods.to(torch.float64))
    cos, sin = torch.cos(ang).to(x.dtype), torch.sin(ang).to(x.dtype)
    xh = x.reshape(B, heads, D, H * W).permute(0, 1, 3, 2)            # b n (hw) d
    x1, x2 = xh[..., : D // 2], xh[..., D // 2:]
    rot = torch.cat([-x2, x1], dim=-1)
    out = xh * cos + rot * sin
    return out.permute(0, 1, 3, 2).reshape(B, C, H, W)


# --------------------------------------------------------------------------------------------
# a2: image encoder  (src/model/naf.py:31-52)
# --------------------------------------------------------------------------------------------
def image_encoder(image: Tensor, out_size: Tuple[int, int], p: Dict[str, Tensor], heads_rope: int) -> Tensor:
    ho, wo = int(out_size[0]), int(out_size[1])
    x = image
    if x.shape[-2] > 4 * ho or x.shape[-1] > 4 * wo:                   # naf.py:39-48
        x = F.interpolate(x, size=(min(x.shape[-2], 4 * ho, 4 * wo), min(x.shape[-1], 4 * wo, 4 * ho)),
                          mode="bilinear", align_corners=False)
    x = conv_stem(x, p)
    x = F.adaptive_avg_pool2d(x, output_size=(ho, wo))                  # naf.py:34
    return rope(x, p["image_encoder.rope.periods"], heads_rope)         # naf.py:51


def key_pool(x: Tensor, lr_size: Tuple[int, int]) -> Tensor:
    """KeyEncoder: adaptive_avg_pool2d of the RoPE'd guidance to the feature grid -- naf.py:63-69."""
    return F.adaptive_avg_pool2d(x, output_size=(int(lr_size[0]), int(lr_size[1])))


# --------------------------------------------------------------------------------------------
# a8: neighbourhood semantics (NATTEN <= 0.17 ``get_window_start``; source not on disk -- restated
# from the published implementation, natten/csrc/.../natten_commons: see module docstring)
# --------------------------------------------------------------------------------------------
def natten_window_start(i: int, L: int, k: int, dil: int) -> int:
    r = k // 2
    if dil <= 1:
        return max(i - r, 0) + (L - i - r - 1 if i + r >= L else 0)
    ni = i - r * dil
    if ni < 0:
        return i % dil
    if i + r * dil >= L:
        m = i % dil
        a = (L // dil) * dil
        b = L - a
        if m < b:
            return L - b + m - 2 * r * dil
        return a + m - k * dil
    return ni


def nearest_exact_src(L_out: int, L_in: int) -> np.ndarray:
    """F.interpolate(mode='nearest-exact') source index: floor((i + 0.5) * L_in / L_out), clamped
    (attentions.py:48-49)."""
    # ATen device arithmetic (UpSample.cuh nearest_neighbor_exact_compute_source_index): all fp32.
    # torch's CPU build agrees except at exact ties (i + 0.5) * L_in / L_out == integer for a few
    # non-integer ratios, where its FMA-contracted code rounds the other way.
    i = np.arange(L_out, dtype=np.float32)
    scale = np.float32(L_in) / np.float32(L_out)
    prod = (i + np.float32(0.5)) * scale
    return np.minimum(np.floor(prod).astype(np.int64), L_in - 1)


def axis_index_table(L_out: int, L_in: int, k: int) -> np.ndarray:
    """[L_out, k] low-res indices a hi-res query at i attends to along one axis: dilated NATTEN
    neighbourhood on the nearest-exact-upsampled grid (attentions.py:54-61), composed with the
    upsampling map.  Raises like NATTEN when k * dilation > L_out."""
    dil = L_out // L_in
    if dil < 1:
        raise ValueError(f"output size {L_out} smaller than feature size {L_in}: dilation would be 0")
    if k * dil > L_out:
        raise ValueError(f"kernel_size * dilation = {k}*{dil} exceeds the axis length {L_out}")
    src = nearest_exact_src(L_out, L_in)
    tab = np.empty((L_out, k), dtype=np.int64)
    for i in range(L_out):
        s = natten_window_start(i, L_out, k, dil)
        tab[i] = src[s + dil * np.arange(k)]
    return tab


def lowres_window_table(L_out: int, L_in: int, k: int) -> np.ndarray:
    """Integer-ratio closed form: pixel i in cell p = i // d attends to cells
    [clamp(p - k//2, 0, L_in - k), +k).  Equal to ``axis_index_table`` when L_out == d * L_in
    (tests/test_oracle.py::test_lowres_form_equals_dilated_form)."""
    assert L_out % L_in == 0 and k <= L_in
    d = L_out // L_in
    p = np.arange(L_out) // d
    s = np.clip(p - k // 2, 0, L_in - k)
    return s[:, None] + np.arange(k)[None, :]


# --------------------------------------------------------------------------------------------
# a7/a8: cross-scale neighbourhood attention  (attentions.py:16-29, 46-61, 75)
# --------------------------------------------------------------------------------------------
def xna_tables(q: Tensor, k_lr: Tensor, v_lr: Tensor, idx_y: np.ndarray, idx_x: np.ndarray, heads: int,
               scale: Optional[float] = None, return_logits: bool = False, rows_per_chunk: int = 8):
    """General form driven by per-axis index tables.

    q [B, Cq, Ho, Wo], k_lr [B, Cq, h, w], v_lr [B, C, h, w]; head g = channels [g*D, (g+1)*D)
    (attentions.py:50,59).  logits[b,g,i,j,ty*kx+tx] = scale * <q[b,g,:,i,j], k[b,g,:,idx_y[i,ty],
    idx_x[j,tx]]> (attentions.py:20-21); softmax over the last dim (:23); weighted sum of v (:24).
    Returns out [B, C, Ho, Wo] (and scaled pre-softmax logits [B, heads, Ho, Wo, ky*kx], which is
    what the reference's return_weights hands back -- attentions.py:27-28).
    """
    B, Cq, Ho, Wo = q.shape
    C = v_lr.shape[1]
    Dq, Dv = Cq // heads, C // heads
    ky, kx = idx_y.shape[1], idx_x.shape[1]
    if scale is None:
        scale = Dq ** -0.5                                              # attentions.py:46
    iy = torch.from_numpy(np.ascontiguousarray(idx_y)).to(q.device)
    ix = torch.from_numpy(np.ascontiguousarray(idx_x)).to(q.device)
    kh = k_lr.reshape(B, heads, Dq, *k_lr.shape[-2:]).permute(0, 1, 3, 4, 2)   # b n h w d
    vh = v_lr.reshape(B, heads, Dv, *v_lr.shape[-2:]).permute(0, 1, 3, 4, 2).to(q.dtype)
    kh = kh.to(q.dtype)
    qh = q.reshape(B, heads, Dq, Ho, Wo).permute(0, 1, 3, 4, 2)                 # b n H W d
    out = torch.empty(B, heads, Ho, Wo, Dv, dtype=q.dtype, device=q.device)
    logits_all = torch.empty(B, heads, Ho, Wo, ky * kx, dtype=q.dtype, device=q.device) if return_logits else None
    for r0 in range(0, Ho, rows_per_chunk):
        r1 = min(Ho, r0 + rows_per_chunk)
        yy = iy[r0:r1]                                                  # [R, ky]
        kg = kh[:, :, yy][:, :, :, :, ix]                               # b n R ky W kx d
        vg = vh[:, :, yy][:, :, :, :, ix]
        kg = kg.permute(0, 1, 2, 4, 3, 5, 6).reshape(B, heads, r1 - r0, Wo, ky * kx, Dq)
        vg = vg.permute(0, 1, 2, 4, 3, 5, 6).reshape(B, heads, r1 - r0, Wo, ky * kx, Dv)
        lg = torch.einsum("bnrwd,bnrwkd->bnrwk", qh[:, :, r0:r1], kg) * scale
        pr = lg.softmax(dim=-1)
        out[:, :, r0:r1] = torch.einsum("bnrwk,bnrwkd->bnrwd", pr, vg)
        if return_logits:
            logits_all[:, :, r0:r1] = lg
    out = out.permute(0, 1, 4, 2, 3).reshape(B, C, Ho, Wo)              # attentions.py:75
    return (out, logits_all) if return_logits else out


def xna(q: Tensor, k_lr: Tensor, v_lr: Tensor, kernel_size, heads: int, return_logits: bool = False):
    """CrossAttention.forward (attentions.py:53-75) with the NATTEN semantics of a8."""
    ky, kx = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
    Ho, Wo = q.shape[-2:]
    h, w = k_lr.shape[-2:]
    return xna_tables(q, k_lr, v_lr, axis_index_table(Ho, h, ky), axis_index_table(Wo, w, kx), heads,
                      return_logits=return_logits)


def xna_lowres(q: Tensor, k_lr: Tensor, v_lr: Tensor, kernel_size: int, heads: int) -> Tensor:
    """Integer-ratio fast restatement: every query of a cell shares the cell's clamped k x k window of
    low-res keys, so the work is one [d*d, D] x [D, k*k] and one [d*d, k*k] x [k*k, Dv] product per
    (cell, head).  Used for the bench's cpu_baseline leg and for large parity cases."""
    B, Cq, Ho, Wo = q.shape
    C = v_lr.shape[1]
    h, w = k_lr.shape[-2:]
    assert Ho % h == 0 and Wo % w == 0
    dy, dx = Ho // h, Wo // w
    k = kernel_size
    Dq, Dv = Cq // heads, C // heads
    sy = torch.clamp(torch.arange(h, device=q.device) - k // 2, 0, h - k)
    sx = torch.clamp(torch.arange(w, device=q.device) - k // 2, 0, w - k)
    wy = sy[:, None] + torch.arange(k, device=q.device)[None, :]       # [h, k]
    wx = sx[:, None] + torch.arange(k, device=q.device)[None, :]       # [w, k]
    kh = k_lr.reshape(B, heads, Dq, h, w).to(q.dtype)
    vh = v_lr.reshape(B, heads, Dv, h, w).to(q.dtype)
    kw = kh[:, :, :, wy][:, :, :, :, :, wx]                            # b n D h k w k
    vw = vh[:, :, :, wy][:, :, :, :, :, wx]
    kw = kw.permute(0, 1, 3, 5, 4, 6, 2).reshape(B, heads, h, w, k * k, Dq)
    vw = vw.permute(0, 1, 3, 5, 4, 6, 2).reshape(B, heads, h, w, k * k, Dv)
    qc = q.reshape(B, heads, Dq, h, dy, w, dx).permute(0, 1, 3, 5, 4, 6, 2).reshape(B, heads, h, w, dy * dx, Dq)
    lg = torch.matmul(qc, kw.transpose(-1, -2)) * (Dq ** -0.5)
    o = torch.matmul(lg.softmax(dim=-1), vw)                           # b n h w (dy dx) Dv
    o = o.reshape(B, heads, h, w, dy, dx, Dv).permute(0, 1, 6, 2, 4, 3, 5)
    return o.reshape(B, C, Ho, Wo)


# --------------------------------------------------------------------------------------------
# a1: the whole forward  (src/model/naf.py:104-116)
# --------------------------------------------------------------------------------------------
def naf_forward(p: Dict[str, Tensor], image: Tensor, features: Tensor, output_size, *, kernel_size=9,
                heads_attn: int = 4, heads_rope: int = 4, return_weights: bool = False):
    out_size = (int(output_size[0]), int(output_size[1]))
    x = image_encoder(image, out_size, p, heads_rope)
    q = x                                                               # naf.py:107 (identity)
    k = key_pool(x, features.shape[-2:])                                # naf.py:108
    return xna(q, k, features, kernel_size, heads_attn, return_logits=return_weights)


def naf_forward_fast(p: Dict[str, Tensor], image: Tensor, features: Tensor, output_size, *, kernel_size=9,
                     heads_attn: int = 4, heads_rope: int = 4) -> Tensor:
    """Same as ``naf_forward`` for integer ratios, through ``xna_lowres`` (cpu_baseline leg)."""
    out_size = (int(output_size[0]), int(output_size[1]))
    x = image_encoder(image, out_size, p, heads_rope)
    k = key_pool(x, features.shape[-2:])
    return xna_lowres(x, k, features, kernel_size, heads_attn)


# --------------------------------------------------------------------------------------------
# backward of a8 (what autograd computes through attentions.py:16-29 + :60-61 in train.py:127-137)
# --------------------------------------------------------------------------------------------
def xna_backward(q: Tensor, k_lr: Tensor, v_lr: Tensor, dout: Tensor, kernel_size: int, heads: int):
    """(dq, dk_lr, dv_lr) of ``xna(q, k_lr, v_lr)`` for the output gradient ``dout``: torch autograd through the
    restated forward (fp64 for a clean reference).  The forward restatement is pinned against the imported
    reference (tests/golden); its derivative is exact calculus, not a second restatement."""
    qd, kd, vd = (t.detach().to(torch.float64).requires_grad_(True) for t in (q, k_lr, v_lr))
    out = xna(qd, kd, vd, kernel_size, heads)
    out.backward(dout.to(torch.float64))
    return qd.grad.to(torch.float32), kd.grad.to(torch.float32), vd.grad.to(torch.float32)
