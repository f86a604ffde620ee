"""Host-side operators over libnaf_hip.so: thin, typed wrappers that hand raw device pointers, element
strides and torch's current HIP stream to the C ABI.  torch is plumbing here (device memory, streams);
every kernel is the library's.  There is no CPU / eager fallback: CPU tensors raise.

Reference counterparts (paths relative to the reference repo):
  axis_index_table   NATTEN neighbourhood rule + nearest-exact map      src/layers/attentions.py:48-61
  stem_conv0/stem_conv  encoder() conv + GroupNorm + SiLU chain         src/layers/convolutions.py:6-92
  rope_tables        RoPE.create_coordinate / angle, sin, cos           src/layers/rope.py:84-105,137-146
  rope_pool          RoPE.forward rotation + KeyEncoder pooling         src/layers/rope.py:147-174, src/model/naf.py:63-69
  pack_values        CrossAttention._resize layout/dtype part           src/layers/attentions.py:50-51
  xna_forward        legacy_attention / na2d                            src/layers/attentions.py:16-29,72
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import XnaArgs, XnaBwdArgs, RopePoolArgs, StemConv0Args, StemConvArgs, KeyPoolArgs, ForwardArgs, I64x3, I64x4

_DT = {torch.bfloat16: _lib.NAF_BF16, torch.float32: _lib.NAF_F32}

# Optional kernel timer (bench.py): an object with start(name) / stop(name) that records HIP events on
# the CURRENT stream tightly around one C-ABI launch.  None in normal operation.
KERNEL_TIMER = None


class _Timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if KERNEL_TIMER is not None:
            KERNEL_TIMER.start(self.name)

    def __exit__(self, *exc):
        if KERNEL_TIMER is not None:
            KERNEL_TIMER.stop(self.name)
        return False
_PATH = {"auto": _lib.XNA_AUTO, "mfma": _lib.XNA_MFMA, "generic": _lib.XNA_GENERIC, "union": _lib.XNA_UNION, "rows": _lib.XNA_ROWS}
_PATH_NAME = {_lib.XNA_MFMA: "mfma", _lib.XNA_GENERIC: "generic", _lib.XNA_UNION: "union", _lib.XNA_ROWS: "rows"}


def _gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"naf_amd: `{name}` is on {t.device}; the NAF hot path only exists as HIP kernels for a "
                           "ROCm device (no CPU fallback). Move the module and its inputs to 'cuda'.")


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _strides4(t: torch.Tensor, dims) -> I64x4:
    return I64x4(*[int(t.stride(d)) for d in dims])


# ------------------------------------------------------------------------------------------------
def axis_index_table(L_out: int, L_in: int, k: int) -> torch.Tensor:
    """[L_out, k] int32 CPU tensor of low-res indices (host function of the library, no GPU needed)."""
    lib = _lib.load()
    if L_out <= 0 or k <= 0:
        raise ValueError(f"axis_index_table: bad sizes L_out={L_out} k={k}")
    out = torch.empty((int(L_out), int(k)), dtype=torch.int32)
    rc = lib.naf_axis_index_table(C.cast(out.data_ptr(), C.POINTER(C.c_int32)), int(L_out), int(L_in), int(k))
    _lib.check(rc, "naf_axis_index_table")
    return out


_table_cache = {}


def device_index_table(L_out: int, L_in: int, k: int, device) -> torch.Tensor:
    key = (int(L_out), int(L_in), int(k), str(device))
    t = _table_cache.get(key)
    if t is None:
        if len(_table_cache) > 64:
            _table_cache.clear()
        t = axis_index_table(L_out, L_in, k).to(device)
        _table_cache[key] = t
    return t


# ------------------------------------------------------------------------------------------------
STATS_SLOTS = 16   # NAF_STATS_SLOTS of include/naf_hip.h: partial copies of every GroupNorm-sum buffer


def new_stats(B: int, device, lead: Tuple[int, ...] = ()) -> torch.Tensor:
    """Zeroed GroupNorm-sum buffer(s) ``[*lead, STATS_SLOTS, B, 8, 2]`` (f64): a producing workgroup adds into one of the
    ``STATS_SLOTS`` copies, consumers add them up (include/naf_hip.h, "guidance conv stem")."""
    return torch.zeros((*lead, STATS_SLOTS, B, 8, 2), dtype=torch.float64, device=device)


def stats_total(stats: torch.Tensor) -> torch.Tensor:
    """``[..., STATS_SLOTS, B, 8, 2]`` -> the sums ``[..., B, 8, 2]``."""
    return stats.sum(dim=-4)


def stats_from_total(total: torch.Tensor) -> torch.Tensor:
    """Sums ``[B, 8, 2]`` (e.g. computed by torch) as a buffer the stem entries read: copy 0 holds them, the rest is zero."""
    st = new_stats(total.shape[0], total.device)
    st[0] = total
    return st


def _stats_ptr(t: Optional[torch.Tensor], B: int, what: str):
    if t is None:
        return None
    if tuple(t.shape) != (STATS_SLOTS, B, 8, 2) or t.dtype != torch.float64 or not t.is_contiguous():
        raise ValueError(f"{what}: GroupNorm sums must be a contiguous f64 [{STATS_SLOTS}, {B}, 8, 2] tensor (ops.new_stats), "
                         f"got {tuple(t.shape)} {t.dtype}")
    return t.data_ptr()


def pack_conv_weight(weight: torch.Tensor) -> torch.Tensor:
    """``w_packed`` of ``naf_stem_conv_fwd`` for a Conv2d weight ``[oc, ic, k, k]`` (k in {1, 3}, oc == ic): bf16 ``[k*k, oc, ic]``
    -- for the 3x3 layers of the default width (128 channels) holding the elements in the order the kernel's lanes keep them,
    ``[9 taps][4 blocks of 32 oc][8 steps of 16 ic][2 halves of 8 ic][32 oc][8 ic]`` (``naf_stem_weight_index``), otherwise
    plain ``weight.permute(2, 3, 0, 1)``.  The data gradient of a layer is the same kernel on
    ``pack_conv_weight(weight.flip(2, 3).transpose(0, 1))``."""
    oc, ic, k, k2 = weight.shape
    if oc != ic or k != k2 or k not in (1, 3):
        raise ValueError(f"pack_conv_weight: expected [C, C, k, k] with k in {{1, 3}}, got {tuple(weight.shape)}")
    w = weight.detach().permute(2, 3, 0, 1).reshape(k * k, oc, ic)
    if k == 3 and oc == 128:
        # (t, wave, n32, ks, half, e) -> (t, wave, ks, half, n32, e)
        w = w.reshape(9, 4, 32, 8, 2, 8).permute(0, 1, 3, 4, 2, 5).reshape(9, 128, 128)
    return w.contiguous().to(torch.bfloat16)


def unpack_conv_weight(w_packed: torch.Tensor) -> torch.Tensor:
    """Inverse of ``pack_conv_weight``: the Conv2d weight ``[oc, ic, k, k]`` (bf16 values as fp32)."""
    taps, oc, ic = w_packed.shape
    k = {1: 1, 9: 3}[int(taps)]
    w = w_packed.float()
    if k == 3 and oc == 128:
        w = w.reshape(9, 4, 8, 2, 32, 8).permute(0, 1, 4, 2, 3, 5).reshape(9, 128, 128)
    return w.reshape(k, k, oc, ic).permute(2, 3, 0, 1).contiguous()


def _fill_stem_conv0(image, weight, bias, y, stats_out) -> StemConv0Args:
    B, Cin, H, W = image.shape
    Cout = int(weight.shape[0])
    if Cin != 3 or weight.shape[1] != 3 or Cout % 16 or not (16 <= Cout <= 256) or weight.dtype != torch.float32 or not weight.is_contiguous():
        raise ValueError(f"stem_conv0: expected a 3-channel image and an f32 [C,3,k,k] weight with C a multiple of 16 up to 256, "
                         f"got {tuple(image.shape)} / {tuple(weight.shape)}")
    a = StemConv0Args()
    a.channels = Cout
    a.image, a.weight, a.bias = image.data_ptr(), weight.data_ptr(), bias.data_ptr()
    a.y = y.data_ptr() if y is not None else None
    a.stats_out = _stats_ptr(stats_out, B, "stem_conv0")
    a.image_dtype, a.ksize, a.B, a.H, a.W = _DT[image.dtype], int(weight.shape[-1]), B, H, W
    a.image_stride = _strides4(image, (0, 1, 2, 3))
    a.y_stride = I64x3(int(y.stride(0)), int(y.stride(1)), int(y.stride(2))) if y is not None else I64x3(0, 0, 0)
    return a


def stem_conv0(image: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, y: Optional[torch.Tensor],
               stats_out: torch.Tensor) -> None:
    """Conv2d(3 -> 128, k in {1, 3}, reflect) + bias.  image [B,3,H,W] f32/bf16 (any strides); weight f32
    [128,3,k,k]; y: bf16 [B,H,W,128] view (128 channels contiguous) or None (statistics only);
    stats_out: ``new_stats(B, device)`` (f64 [STATS_SLOTS,B,8,2], zeroed)."""
    _gpu(image, "image")
    lib = _lib.load()
    if image.dtype not in _DT:
        image = image.float()
    a = _fill_stem_conv0(image, weight, bias, y, stats_out)
    with torch.cuda.device(image.device), _Timed("stem_conv0"):
        rc = lib.naf_stem_conv0_fwd(C.byref(a), _stream(image))
    _lib.check(rc, "naf_stem_conv0_fwd")


def stem_conv(x: Optional[torch.Tensor], stats_in: torch.Tensor, gn_weight: torch.Tensor, gn_bias: torch.Tensor, eps: float,
              w_packed: torch.Tensor, bias: torch.Tensor, y: torch.Tensor, stats_out: Optional[torch.Tensor],
              first=None, keys=None) -> None:
    """GroupNorm(8,128) -> SiLU -> Conv2d(128 -> 128, k in {1,3}, reflect) + bias on bf16 [B,H,W,128] views.
    w_packed: ``pack_conv_weight(conv.weight)`` (bf16 [k*k, C, C]); stats: f64 [STATS_SLOTS,B,8,2] buffers of ``new_stats`` (stats_out zeroed, or None).
    ``first=(image, conv0_weight, conv0_bias)`` (1x1 layers only, ``x=None``): the input is bf16(conv0(image))
    recomputed on the fly; ``stats_in`` then come from ``stem_conv0(..., y=None, ...)``.
    ``keys=(k_slice, tab_y, tab_x)`` (a branch's LAST layer, ``stats_out=None``): ``naf_stem_conv_keys_fwd`` -- the layer also
    writes its 128 channels of the pooled, RoPE'd keys into the bf16 ``[B, h, w, 128]`` view ``k_slice`` (16 x 16 pixel cells);
    raises when the library does not serve the geometry (``stem_conv_keys_supported``)."""
    lib = _lib.load()
    f0 = None
    if first is not None:
        image, w0, b0 = first
        _gpu(image, "image")
        if image.dtype not in _DT:
            image = image.float()
        f0 = _fill_stem_conv0(image, w0, b0, None, None)
        x = y                                                        # shape / device donor only
    _gpu(x, "x")
    B, H, W, Cc = x.shape
    if Cc % 16 or not (16 <= Cc <= 256) or x.dtype != torch.bfloat16 or x.stride(3) != 1 or y.stride(3) != 1:
        raise ValueError("stem_conv: activations must be bf16 [B,H,W,C] with channels contiguous, C a multiple of 16 up to 256")
    if tuple(w_packed.shape[1:]) != (Cc, Cc):
        raise ValueError(f"stem_conv: packed weight {tuple(w_packed.shape)} does not match {Cc} channels")
    taps = w_packed.shape[0]
    a = StemConvArgs()
    a.x, a.y, a.w_packed, a.bias = (None if first is not None else x.data_ptr()), y.data_ptr(), w_packed.data_ptr(), bias.data_ptr()
    if f0 is not None:
        a.first = C.pointer(f0)
    a.gn_weight, a.gn_bias, a.stats_in = gn_weight.data_ptr(), gn_bias.data_ptr(), _stats_ptr(stats_in, B, "stem_conv")
    a.stats_out = _stats_ptr(stats_out, B, "stem_conv")
    a.ksize = {1: 1, 9: 3}[int(taps)]
    a.channels = Cc
    a.B, a.H, a.W, a.eps = B, H, W, float(eps)
    a.x_stride = I64x3(int(x.stride(0)), int(x.stride(1)), int(x.stride(2)))
    a.y_stride = I64x3(int(y.stride(0)), int(y.stride(1)), int(y.stride(2)))
    if keys is not None:
        kp = _fill_key_pool(keys, x)
        with torch.cuda.device(x.device), _Timed("stem_conv%d_keys" % a.ksize):
            rc = lib.naf_stem_conv_keys_fwd(C.byref(a), C.byref(kp), _stream(x))
        _lib.check(rc, "naf_stem_conv_keys_fwd")
        return
    with torch.cuda.device(x.device), _Timed("stem_conv%d" % a.ksize):
        rc = lib.naf_stem_conv_fwd(C.byref(a), _stream(x))
    _lib.check(rc, "naf_stem_conv_fwd")


def _fill_key_pool(keys, x) -> KeyPoolArgs:
    k_slice, tab_y, tab_x = keys
    _gpu(k_slice, "k_slice")
    if k_slice.dtype != torch.bfloat16 or k_slice.dim() != 4 or k_slice.shape[0] != x.shape[0] or k_slice.shape[3] != 128 or k_slice.stride(3) != 1:
        raise ValueError("stem_conv: keys must be a bf16 [B, h, w, 128] view with channels contiguous")
    if tab_y.dtype != torch.float32 or tab_x.dtype != torch.float32 or tab_y.shape[-1] != 16 or tab_x.shape[-1] != 16:
        raise ValueError("stem_conv: keys need the fp32 RoPE tables of 16 periods (heads of 64 channels)")
    kp = KeyPoolArgs()
    kp.k_lr, kp.tab_y, kp.tab_x = k_slice.data_ptr(), tab_y.data_ptr(), tab_x.data_ptr()
    kp.h, kp.w = int(k_slice.shape[1]), int(k_slice.shape[2])
    kp.k_stride = I64x3(int(k_slice.stride(0)), int(k_slice.stride(1)), int(k_slice.stride(2)))
    return kp


def stem_conv_plain(x: torch.Tensor, w_packed: torch.Tensor, y: torch.Tensor, bias: Optional[torch.Tensor] = None) -> None:
    """y = conv(x) (+ bias): ``naf_stem_conv_fwd`` without GroupNorm / SiLU, bf16 [B,H,W,C] views (any width the stem serves), reflect padding for the
    3x3 kernel.  With ``w_packed = pack_conv_weight(weight.flip(2, 3).transpose(0, 1))`` it is a layer's data gradient (see include/naf_hip.h)."""
    lib = _lib.load()
    _gpu(x, "x")
    B, H, W, Cc = x.shape
    if Cc % 16 or not (16 <= Cc <= 256) or x.dtype != torch.bfloat16 or y.dtype != torch.bfloat16 or x.stride(3) != 1 or y.stride(3) != 1:
        raise ValueError("stem_conv_plain: bf16 [B,H,W,C] activations with channels contiguous, C a multiple of 16 up to 256")
    if tuple(w_packed.shape[1:]) != (Cc, Cc) or w_packed.dtype != torch.bfloat16 or not w_packed.is_contiguous():
        raise ValueError(f"stem_conv_plain: packed weight {tuple(w_packed.shape)} / {w_packed.dtype}")
    a = StemConvArgs()
    a.x, a.y, a.w_packed = x.data_ptr(), y.data_ptr(), w_packed.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.gn_weight = a.gn_bias = a.stats_in = a.stats_out = None
    a.ksize = {1: 1, 9: 3}[int(w_packed.shape[0])]
    a.channels = Cc
    a.B, a.H, a.W, a.eps = B, H, W, 0.0
    a.x_stride = I64x3(int(x.stride(0)), int(x.stride(1)), int(x.stride(2)))
    a.y_stride = I64x3(int(y.stride(0)), int(y.stride(1)), int(y.stride(2)))
    with torch.cuda.device(x.device), _Timed("stem_dgrad%d" % a.ksize):
        rc = lib.naf_stem_conv_fwd(C.byref(a), _stream(x))
    _lib.check(rc, "naf_stem_conv_fwd")


def stem_act(x: torch.Tensor, stats_in: torch.Tensor, gn_weight: torch.Tensor, gn_bias: torch.Tensor, eps: float,
             pad: int = 0) -> torch.Tensor:
    """SiLU(GroupNorm(8, C)(x)) as bf16 [B, H + 2 pad, W + 2 pad, C] with a reflected border (``naf_stem_act_fwd``)."""
    lib = _lib.load()
    _gpu(x, "x")
    B, H, W, Cc = x.shape
    out = torch.empty((B, H + 2 * pad, W + 2 * pad, Cc), dtype=torch.bfloat16, device=x.device)
    a = _lib.StemActArgs()
    a.x, a.a, a.gn_weight, a.gn_bias, a.stats_in = x.data_ptr(), out.data_ptr(), gn_weight.data_ptr(), gn_bias.data_ptr(), _stats_ptr(stats_in, B, "stem_act")
    a.B, a.H, a.W, a.channels, a.pad, a.eps = B, H, W, Cc, int(pad), float(eps)
    a.x_stride = I64x3(int(x.stride(0)), int(x.stride(1)), int(x.stride(2)))
    a.a_stride = I64x3(int(out.stride(0)), int(out.stride(1)), int(out.stride(2)))
    with torch.cuda.device(x.device), _Timed("stem_act"):
        rc = lib.naf_stem_act_fwd(C.byref(a), _stream(x))
    _lib.check(rc, "naf_stem_act_fwd")
    return out


def stem_wgrad(dy: torch.Tensor, x: torch.Tensor, stats_in: Optional[torch.Tensor], gn_weight: Optional[torch.Tensor],
               gn_bias: Optional[torch.Tensor], eps: float, ksize: int, with_bias: bool = False, out: Optional[torch.Tensor] = None):
    """Weight gradient [C, C, k, k] (fp32) of y = conv_k(SiLU(GroupNorm(x))) + b: ``naf_stem_wgrad``; dy, x bf16 [B,H,W,C]
    (C = 128: the hand-scheduled kernel of stem_wgrad.hip; other multiples of 16 up to 256: stem_generic_bwd.hip).
    ``stats_in=None``: x already is the activation SiLU(GroupNorm(.)) (``stem_act(..., pad=0)``).  ``with_bias``: also returns
    the bias gradient [128] (sum of dy over pixels, accumulated by the same kernel).  ``out``: a ZEROED fp32 buffer of
    ``k*k*C*C + C`` elements to accumulate into (a training step zeroes one buffer for all its layers) instead of a new one."""
    lib = _lib.load()
    _gpu(x, "x")
    B, H, W, Cc = x.shape
    if (Cc % 16 or not (16 <= Cc <= 256) or tuple(dy.shape) != (B, H, W, Cc) or dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16
            or dy.stride(3) != 1 or x.stride(3) != 1):
        raise ValueError("stem_wgrad: bf16 [B,H,W,C] tensors with channels contiguous, C a multiple of 16 up to 256")
    n_out = ksize * ksize * Cc * Cc + Cc
    if out is not None:
        if out.dtype != torch.float32 or out.numel() != n_out or not out.is_contiguous() or out.device != x.device:
            raise ValueError(f"stem_wgrad: out must be a contiguous zeroed f32 buffer of {n_out} elements")
        buf = out.view(-1)
    else:
        buf = torch.zeros((n_out,), dtype=torch.float32, device=x.device)   # one memset for both
    dw = buf[: ksize * ksize * Cc * Cc].view(ksize, ksize, Cc, Cc)                       # taps outermost: coalesced atomics
    db = buf[ksize * ksize * Cc * Cc:]
    a = _lib.StemWgradArgs()
    a.dy, a.x, a.dw = dy.data_ptr(), x.data_ptr(), dw.data_ptr()
    a.db = db.data_ptr() if with_bias else None
    if stats_in is not None:
        a.gn_weight, a.gn_bias, a.stats_in = gn_weight.data_ptr(), gn_bias.data_ptr(), _stats_ptr(stats_in, B, "stem_wgrad")
    else:
        a.gn_weight = a.gn_bias = a.stats_in = None
    a.ksize, a.B, a.H, a.W, a.eps, a.channels = int(ksize), B, H, W, float(eps), Cc
    a.dy_stride = I64x3(int(dy.stride(0)), int(dy.stride(1)), int(dy.stride(2)))
    a.x_stride = I64x3(int(x.stride(0)), int(x.stride(1)), int(x.stride(2)))
    with torch.cuda.device(x.device), _Timed("stem_wgrad%d" % ksize):
        rc = lib.naf_stem_wgrad(C.byref(a), _stream(x))
    _lib.check(rc, "naf_stem_wgrad")
    return (dw.permute(2, 3, 0, 1), db) if with_bias else dw.permute(2, 3, 0, 1)


def stem_conv0_wgrad(dy: torch.Tensor, image: torch.Tensor, ksize: int, out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dW [C, 3, k, k], db [C]) of the first convolution (``naf_stem_conv0_wgrad``): dy bf16 [B,H,W,C], image [B,3,H,W]."""
    lib = _lib.load()
    _gpu(dy, "dy")
    B, H, W, Cc = dy.shape
    if Cc % 16 or not (16 <= Cc <= 256) or dy.dtype != torch.bfloat16 or dy.stride(3) != 1 or tuple(image.shape) != (B, 3, H, W):
        raise ValueError(f"stem_conv0_wgrad: dy {tuple(dy.shape)} / image {tuple(image.shape)}")
    if image.dtype not in _DT:
        image = image.float()
    nt = 3 * ksize * ksize
    if out is not None:      # a ZEROED contiguous f32 buffer of (3 k k + 1) C elements (see stem_wgrad)
        if out.dtype != torch.float32 or out.numel() != (nt + 1) * Cc or not out.is_contiguous() or out.device != dy.device:
            raise ValueError(f"stem_conv0_wgrad: out must be a contiguous zeroed f32 buffer of {(nt + 1) * Cc} elements")
        buf = out.view(-1)
    else:
        buf = torch.zeros(((nt + 1) * Cc,), dtype=torch.float32, device=dy.device)
    a = _lib.StemConv0WgradArgs()
    a.dy, a.image, a.dw, a.db = dy.data_ptr(), image.data_ptr(), buf.data_ptr(), buf[nt * Cc:].data_ptr()
    a.image_dtype, a.ksize, a.B, a.H, a.W, a.channels = _DT[image.dtype], int(ksize), B, H, W, Cc
    a.dy_stride = I64x3(int(dy.stride(0)), int(dy.stride(1)), int(dy.stride(2)))
    a.image_stride = _strides4(image, (0, 1, 2, 3))
    with torch.cuda.device(dy.device), _Timed("stem_conv0_wgrad"):
        rc = lib.naf_stem_conv0_wgrad(C.byref(a), _stream(dy))
    _lib.check(rc, "naf_stem_conv0_wgrad")
    return buf[: nt * Cc].view(3, ksize, ksize, Cc).permute(3, 0, 1, 2), buf[nt * Cc:]


def stem_conv0_dgrad(dy: torch.Tensor, weight: torch.Tensor, dimage: torch.Tensor, accumulate: bool = False) -> None:
    """Gradient of the first convolution Conv2d(3 -> C, k in {1, 3}, reflect) w.r.t. the image (``naf_stem_conv0_dgrad``):
    dy bf16 [B,H,W,C], weight f32 [C,3,k,k] (the parameter), dimage f32 [B,3,H,W] written or -- ``accumulate`` -- added to."""
    lib = _lib.load()
    _gpu(dy, "dy")
    B, H, W, Cc = dy.shape
    k = int(weight.shape[-1])
    if (Cc % 16 or not (16 <= Cc <= 256) or dy.dtype != torch.bfloat16 or dy.stride(3) != 1 or tuple(dimage.shape) != (B, 3, H, W)
            or dimage.dtype != torch.float32 or tuple(weight.shape) != (Cc, 3, k, k) or weight.dtype != torch.float32 or not weight.is_contiguous()):
        raise ValueError(f"stem_conv0_dgrad: dy {tuple(dy.shape)} / weight {tuple(weight.shape)} / dimage {tuple(dimage.shape)} {dimage.dtype}")
    a = _lib.StemConv0DgradArgs()
    a.dy, a.weight, a.dimage = dy.data_ptr(), weight.data_ptr(), dimage.data_ptr()
    a.ksize, a.B, a.H, a.W, a.channels, a.accumulate = k, B, H, W, Cc, int(bool(accumulate))
    a.dy_stride = I64x3(int(dy.stride(0)), int(dy.stride(1)), int(dy.stride(2)))
    a.dimage_stride = _strides4(dimage, (0, 1, 2, 3))
    with torch.cuda.device(dy.device), _Timed("stem_conv0_dgrad"):
        rc = lib.naf_stem_conv0_dgrad(C.byref(a), _stream(dy))
    _lib.check(rc, "naf_stem_conv0_dgrad")


def stem_act_bwd(da: torch.Tensor, x: torch.Tensor, stats_in: torch.Tensor, gn_weight: torch.Tensor, gn_bias: torch.Tensor,
                 eps: float, dx: torch.Tensor, fold: bool = False, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Backward of SiLU(GroupNorm(x)) (``naf_stem_act_bwd``): writes dx (bf16 [B,H,W,C] view) and returns the fp64 sums
    [B, C, 2] = per sample {d gn_bias, d gn_weight}.  ``fold``: da is [B, H+2, W+2, C], the gradient on the reflect-padded
    domain."""
    lib = _lib.load()
    _gpu(x, "x")
    B, H, W, Cc = x.shape
    want = (B, H + 2, W + 2, Cc) if fold else (B, H, W, Cc)
    if tuple(da.shape) != want or da.dtype != torch.bfloat16 or da.stride(3) != 1 or dx.stride(3) != 1 or tuple(dx.shape) != (B, H, W, Cc):
        raise ValueError(f"stem_act_bwd: da {tuple(da.shape)} (want {want}) / dx {tuple(dx.shape)}")
    if sums is not None:     # a ZEROED contiguous fp64 [B, C, 2] to accumulate into
        if sums.dtype != torch.float64 or tuple(sums.shape) != (B, Cc, 2) or not sums.is_contiguous() or sums.device != x.device:
            raise ValueError(f"stem_act_bwd: sums must be a contiguous zeroed f64 [{B}, {Cc}, 2]")
    else:
        sums = torch.zeros((B, Cc, 2), dtype=torch.float64, device=x.device)
    a = _lib.StemActBwdArgs()
    a.da, a.x, a.dx = da.data_ptr(), x.data_ptr(), dx.data_ptr()
    a.gn_weight, a.gn_bias, a.stats_in, a.sums = gn_weight.data_ptr(), gn_bias.data_ptr(), _stats_ptr(stats_in, B, "stem_act_bwd"), sums.data_ptr()
    a.B, a.H, a.W, a.channels, a.fold, a.phase, a.eps = B, H, W, Cc, int(bool(fold)), 0, float(eps)
    a.da_stride = I64x3(int(da.stride(0)), int(da.stride(1)), int(da.stride(2)))
    a.x_stride = I64x3(int(x.stride(0)), int(x.stride(1)), int(x.stride(2)))
    a.dx_stride = I64x3(int(dx.stride(0)), int(dx.stride(1)), int(dx.stride(2)))
    with torch.cuda.device(x.device), _Timed("stem_act_bwd"):
        rc = lib.naf_stem_act_bwd(C.byref(a), _stream(x))
    _lib.check(rc, "naf_stem_act_bwd")
    return sums


# ------------------------------------------------------------------------------------------------
def rope_tables(periods: torch.Tensor, Ho: int, Wo: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin tables [Ho, 2, P] and [Wo, 2, P] (fp32) for the module's `periods` buffer [P]."""
    _gpu(periods, "periods")
    lib = _lib.load()
    per = periods.detach().to(torch.float32).contiguous()
    P = per.numel()
    ty = torch.empty((Ho, 2, P), dtype=torch.float32, device=per.device)
    tx = torch.empty((Wo, 2, P), dtype=torch.float32, device=per.device)
    with torch.cuda.device(per.device):
        rc = lib.naf_rope_tables(ty.data_ptr(), tx.data_ptr(), per.data_ptr(), P, int(Ho), int(Wo), _stream(per))
    _lib.check(rc, "naf_rope_tables")
    return ty, tx


def rope_pool(x: torch.Tensor, tab_y: torch.Tensor, tab_x: torch.Tensor, heads: int, lr_size,
              q_layout: str = "head_major", write_q: bool = True) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """x: logical [B, Cq, Ho, Wo] (bf16/fp32, any strides; channels-last is the fast layout).
    Returns q [B, heads, Ho, Wo, Dh] bf16 (RoPE'd queries) and k_lr [B, heads, h, w, Dh] bf16
    (adaptive-avg-pooled RoPE'd keys), both as 5-D strided views with Dh contiguous.
    ``write_q=False``: keys only (q is None) -- for ``xna_forward(..., rope_tables=...)`` which rotates the
    queries as it loads them."""
    _gpu(x, "guidance features")
    lib = _lib.load()
    if x.dtype not in _DT:
        x = x.float()
    B, Cq, Ho, Wo = x.shape
    h, w = int(lr_size[0]), int(lr_size[1])
    if Cq % heads:
        raise ValueError(f"rope_pool: {Cq} channels not divisible by {heads} heads")
    Dh = Cq // heads
    dev = x.device
    if not write_q:
        q = None
    elif q_layout == "head_major":
        q = torch.empty((B, heads, Ho, Wo, Dh), dtype=torch.bfloat16, device=dev)
    elif q_layout == "channels_last":
        q = torch.empty((B, Ho, Wo, heads, Dh), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
    else:
        raise ValueError(f"unknown q_layout {q_layout!r}")
    k = torch.empty((B, h, w, heads, Dh), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
    a = RopePoolArgs()
    a.x, a.q, a.k_lr = x.data_ptr(), (q.data_ptr() if q is not None else None), k.data_ptr()
    a.tab_y, a.tab_x = tab_y.data_ptr(), tab_x.data_ptr()
    a.x_dtype = _DT[x.dtype]
    a.B, a.Cq, a.heads, a.Ho, a.Wo, a.h, a.w = B, Cq, heads, Ho, Wo, h, w
    a.x_stride = _strides4(x, (0, 1, 2, 3))
    a.q_stride = _strides4(q, (0, 1, 2, 3)) if q is not None else I64x4(0, 0, 0, 0)
    a.k_stride = _strides4(k, (0, 1, 2, 3))
    if tab_y.shape != (Ho, 2, Dh // 4) or tab_x.shape != (Wo, 2, Dh // 4):
        raise ValueError(f"rope_pool: tables {tuple(tab_y.shape)}/{tuple(tab_x.shape)} do not match Ho={Ho} Wo={Wo} Dh/4={Dh // 4}")
    with torch.cuda.device(dev), _Timed("rope_pool"):
        rc = lib.naf_rope_pool_fwd(C.byref(a), _stream(x))
    _lib.check(rc, "naf_rope_pool_fwd")
    return q, k


def preshrink_image(image: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(image, size, mode="bilinear", align_corners=False) of naf.py:39-48 -> fp32 [B, 3, Hs, Ws]."""
    _gpu(image, "image")
    if image.dtype not in _DT or image.dim() != 4 or image.shape[1] != 3:
        raise ValueError("preshrink_image: expected a [B, 3, H, W] float32 / bfloat16 image")
    B, _, H, W = image.shape
    Hs, Ws = int(size[0]), int(size[1])
    out = torch.empty((B, 3, Hs, Ws), dtype=torch.float32, device=image.device)
    st = _strides4(image, (0, 1, 2, 3))
    lib = _lib.load()
    with torch.cuda.device(image.device), _Timed("preshrink"):
        rc = lib.naf_preshrink_image(out.data_ptr(), image.data_ptr(), _DT[image.dtype], B, H, W, Hs, Ws, C.byref(st), _stream(image))
    _lib.check(rc, "naf_preshrink_image")
    return out


def pool_guidance(x: torch.Tensor, output_size) -> torch.Tensor:
    """adaptive_avg_pool2d of the bf16 channels-last guidance [B, C, H, W] (logical) to ``output_size`` (naf.py:34);
    returns a logical [B, C, Ho, Wo] view of a dense channels-last buffer."""
    _gpu(x, "x")
    B, Cc, H, W = x.shape
    Ho, Wo = int(output_size[0]), int(output_size[1])
    xc = x.permute(0, 2, 3, 1)
    if x.dtype != torch.bfloat16 or not xc.is_contiguous() or Cc % 8:
        raise ValueError("pool_guidance: expected a dense channels-last bf16 tensor with C % 8 == 0")
    y = torch.empty((B, Ho, Wo, Cc), dtype=torch.bfloat16, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device), _Timed("pool_guidance"):
        rc = lib.naf_pool_guidance(y.data_ptr(), xc.data_ptr(), B, H, W, Ho, Wo, Cc, _stream(x))
    _lib.check(rc, "naf_pool_guidance")
    return y.permute(0, 3, 1, 2)


def pack_values(v: torch.Tensor) -> torch.Tensor:
    """[B, C, h, w] (bf16/fp32, any strides) -> dense channels-last bf16 [B, h, w, C]."""
    _gpu(v, "lr_features")
    lib = _lib.load()
    if v.dtype not in _DT:
        v = v.float()
    B, Cc, h, w = v.shape
    vp = torch.empty((B, h, w, Cc), dtype=torch.bfloat16, device=v.device)
    st = _strides4(v, (0, 1, 2, 3))
    with torch.cuda.device(v.device):
        rc = lib.naf_pack_values(vp.data_ptr(), v.data_ptr(), _DT[v.dtype], B, Cc, h, w, st, _stream(v))
    _lib.check(rc, "naf_pack_values")
    return vp


def _fill_xna(q, k, v, out, logits, idx_y, idx_x, ky, kx, path, scale, rope_tables=None) -> XnaArgs:
    B, heads, Ho, Wo, Dq = q.shape
    _, _, h, w, Dv = v.shape
    a = XnaArgs()
    a.q, a.k_lr, a.v_lr, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.logits = logits.data_ptr() if logits is not None else None
    a.idx_y = idx_y.data_ptr() if idx_y is not None else None
    a.idx_x = idx_x.data_ptr() if idx_x is not None else None
    if rope_tables is not None:
        ty, tx = rope_tables
        if ty.dtype != torch.float32 or tx.dtype != torch.float32 or tuple(ty.shape) != (q.shape[2], 2, q.shape[4] // 4) \
                or tuple(tx.shape) != (q.shape[3], 2, q.shape[4] // 4) or not (ty.is_contiguous() and tx.is_contiguous()):
            raise ValueError("xna: rope_tables must be the fp32 [Ho,2,Dq/4] / [Wo,2,Dq/4] pair from ops.rope_tables")
        a.rope_tab_y, a.rope_tab_x = ty.data_ptr(), tx.data_ptr()
    a.B, a.heads, a.Ho, a.Wo, a.h, a.w, a.Dq, a.Dv, a.ky, a.kx = B, heads, Ho, Wo, h, w, Dq, Dv, ky, kx
    a.out_dtype = _DT[out.dtype]
    a.path = _PATH[path]
    a.scale = float(scale) if scale else 0.0
    a.q_stride = _strides4(q, (0, 1, 2, 3))
    a.k_stride = _strides4(k, (0, 1, 2, 3))
    a.v_stride = _strides4(v, (0, 1, 2, 3))
    a.o_stride = _strides4(out, (0, 1, 2, 3))
    return a


def xna_forward(q: torch.Tensor, k_lr: torch.Tensor, v_lr: torch.Tensor, kernel_size, *,
                out_dtype: torch.dtype = torch.bfloat16, return_logits: bool = False, path: str = "auto",
                scale: Optional[float] = None, out: Optional[torch.Tensor] = None, rope_tables=None):
    """Cross-scale neighbourhood attention forward.

    ``rope_tables=(tab_y, tab_x)``: ``q`` is the UN-rotated guidance and the kernel applies RoPE while loading
    it (see ``xna_rope_fusable``; raises NafHipError when the shapes do not allow it).

    q [B, heads, Ho, Wo, Dq] bf16, k_lr [B, heads, h, w, Dq] bf16, v_lr [B, heads, h, w, Dv] bf16 --
    5-D strided views with the last dim contiguous.  Returns ``out`` as a [B, heads, Ho, Wo, Dv] view
    of a dense channels-last [B, Ho, Wo, heads*Dv] buffer (and, with ``return_logits``, the scaled
    pre-softmax scores [B, heads, Ho, Wo, ky*kx] fp32 -- what the reference's return_weights gives).
    """
    for t, n in ((q, "q"), (k_lr, "k_lr"), (v_lr, "v_lr")):
        _gpu(t, n)
        if t.dtype != torch.bfloat16:
            raise TypeError(f"xna_forward: {n} must be bfloat16, got {t.dtype}")
        if t.dim() != 5 or t.stride(4) != 1:
            raise ValueError(f"xna_forward: {n} must be 5-D [B, heads, H, W, D] with D contiguous")
    lib = _lib.load()
    ky, kx = (int(kernel_size), int(kernel_size)) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
    B, heads, Ho, Wo, Dq = q.shape
    _, _, h, w, Dv = v_lr.shape
    if k_lr.shape != (B, heads, h, w, Dq):
        raise ValueError(f"xna_forward: k_lr shape {tuple(k_lr.shape)} does not match q/v")
    if out_dtype not in _DT:
        raise TypeError(f"xna_forward: out_dtype {out_dtype} not supported (bfloat16 / float32)")
    dev = q.device
    if out is None:
        out = torch.empty((B, Ho, Wo, heads, Dv), dtype=out_dtype, device=dev).permute(0, 3, 1, 2, 4)
    logits = torch.empty((B, heads, Ho, Wo, ky * kx), dtype=torch.float32, device=dev) if return_logits else None
    a = _fill_xna(q, k_lr, v_lr, out, logits, None, None, ky, kx, path, scale, rope_tables)
    sel = lib.naf_xna_select(C.byref(a))
    if sel < 0:
        _lib.check(-sel, "naf_xna_select")
    if sel != _lib.XNA_MFMA:   # the table-driven kernels (MFMA "union" and generic) read the canonical index tables
        iy = device_index_table(Ho, h, ky, dev)
        ix = device_index_table(Wo, w, kx, dev)
        a.idx_y, a.idx_x = iy.data_ptr(), ix.data_ptr()
    with torch.cuda.device(dev), _Timed("xna_" + _PATH_NAME[sel]):
        rc = lib.naf_xna_fwd(C.byref(a), _stream(q))
    _lib.check(rc, "naf_xna_fwd")
    return (out, logits) if return_logits else out


def _fill_xna_bwd(q, k, v, dout, dq, dk, dv, ky, kx, scale) -> XnaBwdArgs:
    B, heads, Ho, Wo, Dq = q.shape
    _, _, h, w, Dv = v.shape
    a = XnaBwdArgs()
    a.q, a.k_lr, a.v_lr, a.dout, a.dq = q.data_ptr(), k.data_ptr(), v.data_ptr(), dout.data_ptr(), dq.data_ptr()
    a.dk_lr, a.dv_lr = dk.data_ptr(), dv.data_ptr()
    a.B, a.heads, a.Ho, a.Wo, a.h, a.w, a.Dq, a.Dv, a.ky, a.kx = B, heads, Ho, Wo, h, w, Dq, Dv, ky, kx
    a.scale = float(scale) if scale else 0.0
    a.q_stride, a.k_stride, a.v_stride = _strides4(q, (0, 1, 2, 3)), _strides4(k, (0, 1, 2, 3)), _strides4(v, (0, 1, 2, 3))
    a.dout_stride, a.dq_stride = _strides4(dout, (0, 1, 2, 3)), _strides4(dq, (0, 1, 2, 3))
    return a


_BWD_PATHS = {"auto": _lib.XNA_AUTO, "mfma": _lib.XNA_MFMA, "rows": _lib.XNA_ROWS, "generic": _lib.XNA_GENERIC}


def xna_backward_supported(q: torch.Tensor, k_lr: torch.Tensor, v_lr: torch.Tensor, kernel_size) -> bool:
    """True when ``xna_backward`` runs the MFMA cell kernel for these shapes (otherwise: the table-driven one)."""
    lib = _lib.load()
    ky, kx = (int(kernel_size), int(kernel_size)) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
    a = _fill_xna_bwd(q, k_lr, v_lr, q, q, q, q, ky, kx, None)      # shape / alignment query only
    B, heads, Ho, Wo, Dq = q.shape
    Dv = v_lr.shape[-1]
    a.dout_stride = I64x4(Ho * Wo * heads * Dv, Dv, Wo * heads * Dv, heads * Dv)
    a.dq_stride = I64x4(Ho * Wo * heads * Dq, Dq, Wo * heads * Dq, heads * Dq)
    return lib.naf_xna_bwd_supported(C.byref(a)) == _lib.XNA_MFMA


def xna_backward_select(q: torch.Tensor, k_lr: torch.Tensor, v_lr: torch.Tensor, kernel_size) -> str:
    """Which kernel ``xna_backward`` runs for these shapes: "mfma" (cell kernel), "rows" (row-streaming matrix-core kernel: the integer ratios
    the cell kernel does not take -- the reference's denoising call, its own training geometry, patch-14 backbones) or "generic" (table-driven scalar kernel)."""
    lib = _lib.load()
    ky, kx = (int(kernel_size), int(kernel_size)) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
    a = _fill_xna_bwd(q, k_lr, v_lr, q, q, q, q, ky, kx, None)      # shape / alignment query only
    B, heads, Ho, Wo, Dq = q.shape
    Dv = v_lr.shape[-1]
    a.dout_stride = I64x4(Ho * Wo * heads * Dv, Dv, Wo * heads * Dv, heads * Dv)
    a.dq_stride = I64x4(Ho * Wo * heads * Dq, Dq, Wo * heads * Dq, heads * Dq)
    sel = lib.naf_xna_bwd_supported(C.byref(a))
    if sel < 0:
        _lib.check(-sel, "naf_xna_bwd_supported")
    return {_lib.XNA_MFMA: "mfma", _lib.XNA_ROWS: "rows"}.get(sel, "generic")


def xna_backward_chunks(q: torch.Tensor, k_lr: torch.Tensor, v_lr: torch.Tensor, kernel_size) -> list:
    """Channel-chunk widths of the launches the cell backward issues for these shapes (``naf_xna_bwd_chunk_plan``): one entry = the
    whole head in one launch, [] = another kernel serves the call.  dQ carries one bf16 rounding per chunk (include/naf_hip.h)."""
    lib = _lib.load()
    ky, kx = (int(kernel_size), int(kernel_size)) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
    a = _fill_xna_bwd(q, k_lr, v_lr, q, q, q, q, ky, kx, None)      # shape / alignment query only
    B, heads, Ho, Wo, Dq = q.shape
    Dv = v_lr.shape[-1]
    a.dout_stride = I64x4(Ho * Wo * heads * Dv, Dv, Wo * heads * Dv, heads * Dv)
    a.dq_stride = I64x4(Ho * Wo * heads * Dq, Dq, Wo * heads * Dq, heads * Dq)
    out = (C.c_int32 * 16)()
    n = lib.naf_xna_bwd_chunk_plan(C.byref(a), out, 16)
    if n < 0:
        _lib.check(-n, "naf_xna_bwd_chunk_plan")
    return [int(out[i]) for i in range(min(n, 16))]


def xna_backward(q: torch.Tensor, k_lr: torch.Tensor, v_lr: torch.Tensor, dout: torch.Tensor, kernel_size, *,
                 scale: Optional[float] = None, path: str = "auto"):
    """Gradients of ``xna_forward`` w.r.t. q, k_lr, v_lr given ``dout`` (5-D [B, heads, Ho, Wo, Dv], any strides with
    Dv contiguous; cast to bf16).  Returns (dq bf16 [B,heads,Ho,Wo,Dq] view of a channels-last buffer,
    dk_lr fp32 [B,heads,h,w,Dq] view, dv_lr fp32 [B,heads,h,w,Dv] view).  ``path`` (naf_xna_bwd_args.path): "auto", or insist on "mfma"
    (cell kernels), "rows" (row-streaming matrix-core kernel) or "generic" -- the table-driven scalar kernel, which serves EVERY shape and
    is the independent reference of the parity tests (until 0.4.1 "generic" only withheld the row-streaming kernel's workspace: shapes the
    cell kernels take ran the cell kernel again, and the tests that compared the two compared it with itself)."""
    for t, n in ((q, "q"), (k_lr, "k_lr"), (v_lr, "v_lr")):
        _gpu(t, n)
        if t.dtype != torch.bfloat16 or t.dim() != 5 or t.stride(4) != 1:
            raise TypeError(f"xna_backward: {n} must be a bfloat16 5-D [B, heads, H, W, D] view with D contiguous")
    lib = _lib.load()
    ky, kx = (int(kernel_size), int(kernel_size)) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
    B, heads, Ho, Wo, Dq = q.shape
    _, _, h, w, Dv = v_lr.shape
    if tuple(dout.shape) != (B, heads, Ho, Wo, Dv):
        raise ValueError(f"xna_backward: dout shape {tuple(dout.shape)} != {(B, heads, Ho, Wo, Dv)}")
    if dout.dtype != torch.bfloat16 or dout.stride(4) != 1:
        dout = dout.permute(0, 2, 3, 1, 4).to(torch.bfloat16).contiguous().permute(0, 3, 1, 2, 4)
    dev = q.device
    dq = torch.empty((B, Ho, Wo, heads, Dq), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
    dk = torch.zeros((B, h, w, heads, Dq), dtype=torch.float32, device=dev)
    dv = torch.zeros((B, h, w, heads, Dv), dtype=torch.float32, device=dev)
    if path not in _BWD_PATHS:
        raise ValueError(f"xna_backward: path must be one of {sorted(_BWD_PATHS)}, got {path!r}")
    a = _fill_xna_bwd(q, k_lr, v_lr, dout, dq, dk, dv, ky, kx, scale)
    a.path = _BWD_PATHS[path]
    sel = lib.naf_xna_bwd_supported(C.byref(a))
    if sel < 0:
        _lib.check(-sel, "naf_xna_bwd_supported")
    if sel in (_lib.XNA_GENERIC, _lib.XNA_ROWS):
        iy = device_index_table(Ho, h, ky, dev)
        ix = device_index_table(Wo, w, kx, dev)
        a.idx_y, a.idx_x = iy.data_ptr(), ix.data_ptr()
    if sel == _lib.XNA_ROWS:    # matrix-core backward of the denoising shapes: per-query softmax statistics live in a workspace
        ws = torch.empty(int(lib.naf_xna_bwd_workspace_bytes(C.byref(a))), dtype=torch.uint8, device=dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    with torch.cuda.device(dev), _Timed("xna_bwd"):
        rc = lib.naf_xna_bwd(C.byref(a), _stream(q))
    _lib.check(rc, "naf_xna_bwd")
    return dq, dk.permute(0, 3, 1, 2, 4), dv.permute(0, 3, 1, 2, 4)


def rope_pool_bwd(dq: torch.Tensor, dk: torch.Tensor, tab_y: torch.Tensor, tab_x: torch.Tensor, out_size) -> torch.Tensor:
    """Backward of ``rope_pool``: dq bf16 [B, heads, Ho, Wo, Dh], dk [B, heads, h, w, Dh] (cast to fp32) -> dx, a logical
    [B, heads * Dh, Ho, Wo] view of a channels-last bf16 buffer (``naf_rope_pool_bwd``)."""
    lib = _lib.load()
    _gpu(dq, "dq")
    B, heads, Ho, Wo, Dh = dq.shape
    h, w = dk.shape[2], dk.shape[3]
    if dq.dtype != torch.bfloat16 or dq.stride(4) != 1:
        dq = dq.to(torch.bfloat16).contiguous()
    dk = dk.float()
    if dk.stride(4) != 1:
        dk = dk.contiguous()
    dx = torch.empty((B, Ho, Wo, heads * Dh), dtype=torch.bfloat16, device=dq.device).permute(0, 3, 1, 2)
    a = _lib.RopePoolBwdArgs()
    a.dq, a.dk_lr, a.dx, a.tab_y, a.tab_x = dq.data_ptr(), dk.data_ptr(), dx.data_ptr(), tab_y.data_ptr(), tab_x.data_ptr()
    a.B, a.Cq, a.heads, a.Ho, a.Wo, a.h, a.w = B, heads * Dh, heads, Ho, Wo, h, w
    a.dq_stride = _strides4(dq, (0, 1, 2, 3))
    a.dk_stride = _strides4(dk, (0, 1, 2, 3))
    a.dx_stride = _strides4(dx, (0, 1, 2, 3))
    with torch.cuda.device(dq.device), _Timed("rope_pool_bwd"):
        rc = lib.naf_rope_pool_bwd(C.byref(a), _stream(dq))
    _lib.check(rc, "naf_rope_pool_bwd")
    return dx


class RopePoolFunction(torch.autograd.Function):
    """Differentiable ``rope_pool`` (queries and pooled keys of the guidance): forward = naf_rope_pool_fwd, backward =
    naf_rope_pool_bwd.  x: logical [B, Cq, Ho, Wo] bf16, channels-last."""

    @staticmethod
    def forward(ctx, x, tab_y, tab_x, heads, lr_size):
        q5, k5 = rope_pool(x, tab_y, tab_x, heads, lr_size, q_layout="channels_last")
        ctx.save_for_backward(tab_y, tab_x)
        ctx.size = tuple(x.shape[-2:])
        return q5, k5

    @staticmethod
    def backward(ctx, dq, dk):
        tab_y, tab_x = ctx.saved_tensors
        return rope_pool_bwd(dq, dk, tab_y, tab_x, ctx.size), None, None, None, None


class XnaFunction(torch.autograd.Function):
    """Differentiable ``xna_forward``: forward = naf_xna_fwd, backward = naf_xna_bwd (MFMA or table-driven kernels)."""

    @staticmethod
    def forward(ctx, q, k_lr, v_lr, kernel_size, scale, out_dtype, *rest):
        return_logits = bool(rest[0]) if rest else False       # optional seventh argument
        ctx.nrest = len(rest)
        ctx.save_for_backward(q, k_lr, v_lr)
        ctx.kernel_size, ctx.scale = kernel_size, scale
        if return_logits:
            # return_weights on a gradient-enabled call (attentions.py:64-67 works under autograd): the scaled pre-softmax scores of the
            # very q / k this differentiable step uses, as a second output WITHOUT a gradient (nothing in the reference's callers
            # differentiates through them: notebooks/attention_maps.ipynb reads them for display)
            out, logits = xna_forward(q, k_lr, v_lr, kernel_size, out_dtype=out_dtype, path="auto", scale=scale, return_logits=True)
            ctx.mark_non_differentiable(logits)
            return out, logits
        return xna_forward(q, k_lr, v_lr, kernel_size, out_dtype=out_dtype, path="auto", scale=scale)

    @staticmethod
    def backward(ctx, dout, *unused):
        q, k_lr, v_lr = ctx.saved_tensors
        dq, dk, dv = xna_backward(q, k_lr, v_lr, dout, ctx.kernel_size, scale=ctx.scale)
        return (dq, dk.to(k_lr.dtype), dv.to(v_lr.dtype), None, None, None) + (None,) * ctx.nrest


def xna_rope_fusable(q: torch.Tensor, lr_size, Dv: int, kernel_size, rope_tables, out_dtype=torch.bfloat16,
                     path: str = "auto") -> bool:
    """True when ``xna_forward(q, ..., rope_tables=...)`` can rotate the queries on load for these shapes (MFMA path,
    Wo/w a multiple of 16).  ``q``: the un-rotated guidance as a 5-D [B, heads, Ho, Wo, Dq] bf16 view."""
    if q.dtype != torch.bfloat16 or q.dim() != 5 or q.stride(4) != 1 or out_dtype not in _DT:
        return False
    lib = _lib.load()
    ky, kx = (int(kernel_size), int(kernel_size)) if isinstance(kernel_size, int) else tuple(kernel_size)
    B, heads, Ho, Wo, Dq = q.shape
    h, w = int(lr_size[0]), int(lr_size[1])
    a = XnaArgs()
    a.q = a.k_lr = a.v_lr = a.out = q.data_ptr()      # shape / alignment query only: nothing is dereferenced
    a.rope_tab_y, a.rope_tab_x = rope_tables[0].data_ptr(), rope_tables[1].data_ptr()
    a.B, a.heads, a.Ho, a.Wo, a.h, a.w, a.Dq, a.Dv, a.ky, a.kx = B, heads, Ho, Wo, h, w, Dq, int(Dv), ky, kx
    a.out_dtype, a.path, a.scale = _DT[out_dtype], _PATH[path], 0.0
    a.q_stride = _strides4(q, (0, 1, 2, 3))
    a.k_stride = I64x4(h * w * heads * Dq, Dq, w * heads * Dq, heads * Dq)
    a.v_stride = I64x4(h * w * heads * Dv, Dv, w * heads * Dv, heads * Dv)
    a.o_stride = I64x4(Ho * Wo * heads * Dv, Dv, Wo * heads * Dv, heads * Dv)
    return lib.naf_xna_select(C.byref(a)) == _lib.XNA_MFMA


def xna_select(q, k_lr, v_lr, kernel_size, out_dtype=torch.bfloat16, return_logits=False, path="auto") -> str:
    """Name of the kernel ``xna_forward`` would dispatch to ('mfma' / 'generic') -- for tests / bench."""
    lib = _lib.load()
    ky, kx = (int(kernel_size), int(kernel_size)) if isinstance(kernel_size, int) else tuple(kernel_size)
    B, heads, Ho, Wo, Dq = q.shape
    Dv = v_lr.shape[-1]
    out = torch.empty((1, 1, 1, heads, Dv), dtype=out_dtype, device=q.device).permute(0, 3, 1, 2, 4)
    fake = torch.empty((1,), dtype=torch.float32, device=q.device) if return_logits else None
    a = _fill_xna(q, k_lr, v_lr, out, fake, None, None, ky, kx, path, None)
    a.o_stride = I64x4(Ho * Wo * heads * Dv, Dv, Wo * heads * Dv, heads * Dv)
    sel = lib.naf_xna_select(C.byref(a))
    if sel < 0:
        _lib.check(-sel, "naf_xna_select")
    return _PATH_NAME[sel]


# ------------------------------------------------------------------------------------------------
class _AuxPool:
    """The ``naf_forward_aux`` objects (a second stream and the fork / join events) this host LENDS to ``naf_forward_ex``.  The
    library owns none (C ABI 0.4.0): one per (host thread, device, caller stream), so concurrent forwards never share fork / join
    events and a hipGraph capture pulls in a stream nothing else uses.  The sixteen most recently used are kept; a bundle that is
    lent out at the moment (``lease``: inside a ``naf_forward_ex`` call of some thread) is never the one that is destroyed."""
    LIMIT = 16

    def __init__(self):
        self._by_key: Dict[tuple, list] = {}      # key -> [ForwardAux, leases outstanding]
        self._lock = threading.Lock()

    def _entry(self, device_index: int, stream_handle: int) -> list:      # caller holds the lock
        key = (threading.get_ident(), device_index, stream_handle)
        ent = self._by_key.pop(key, None)
        if ent is None:
            lib = _lib.load()
            idle = [k for k, e in self._by_key.items() if e[1] == 0]
            while len(self._by_key) >= self.LIMIT and idle:
                old = self._by_key.pop(idle.pop(0))                         # least recently used idle one (dicts keep insertion order)
                lib.naf_forward_aux_destroy(C.byref(old[0]))                # work already queued on a destroyed stream still completes
            aux = _lib.ForwardAux()
            with torch.cuda.device(device_index):
                _lib.check(lib.naf_forward_aux_create(C.byref(aux)), "naf_forward_aux_create")
            ent = [aux, 0]
        self._by_key[key] = ent
        return ent

    def get(self, device_index: int, stream_handle: int) -> "_lib.ForwardAux":
        with self._lock:
            return self._entry(device_index, stream_handle)[0]

    @contextlib.contextmanager
    def lease(self, device_index: int, stream_handle: int):
        """The bundle of (this thread, device, stream) for the duration of one foreign call."""
        with self._lock:
            ent = self._entry(device_index, stream_handle)
            ent[1] += 1
        try:
            yield ent[0]
        finally:
            with self._lock:
                ent[1] -= 1

    def clear(self) -> None:
        with self._lock:
            lib = _lib.load()
            for k in [k for k, e in self._by_key.items() if e[1] == 0]:
                lib.naf_forward_aux_destroy(C.byref(self._by_key.pop(k)[0]))


AUX_POOL = _AuxPool()


def forward_aux(device, stream=None) -> "_lib.ForwardAux":
    """The second-stream bundle ``ForwardPlan.run`` lends to the library for forwards issued by this thread on ``stream`` (default:
    the current one) of ``device``.  Call it BEFORE a hipGraph capture on that stream so that nothing is created while capturing."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = stream if stream is not None else torch.cuda.current_stream(idx)
    return AUX_POOL.get(idx, int(st.cuda_stream))


class ForwardPlan:
    """Argument block of ``naf_forward`` (the whole forward in ONE foreign call) for fixed parameters and shapes.
    Built once per (parameter versions, shapes); ``run`` only swaps the image / features / output pointers."""

    def __init__(self, branches, nlayer: int, gn_eps: float, tabs, image: torch.Tensor, features: torch.Tensor,
                 heads: int, ksize: int, out_dtype: torch.dtype, scale: Optional[float], output_size=None, heads_rope: int = 0):
        lib = _lib.load()
        B, _, H, W = image.shape
        Ho, Wo = (int(output_size[0]), int(output_size[1])) if output_size is not None else (H, W)
        _, Cc, h, w = features.shape
        a = ForwardArgs()
        a.tab_y, a.tab_x = tabs[0].data_ptr(), tabs[1].data_ptr()
        a.nlayer = nlayer
        a.image_dtype, a.feat_dtype, a.out_dtype = _DT[image.dtype], _DT[features.dtype], _DT[out_dtype]
        a.B, a.H, a.W, a.h, a.w, a.C, a.heads, a.ksize = B, H, W, h, w, Cc, heads, ksize
        a.Ho, a.Wo = Ho, Wo
        a.heads_rope = int(heads_rope)
        a.gn_eps = float(gn_eps)
        a.scale = float(scale) if scale else 0.0
        self._keep = [tabs]
        for br, (w0, b0, k0, kb, layers) in enumerate(branches):
            sb = a.branch[br]
            sb.conv0_weight, sb.conv0_bias, sb.conv0_ksize, sb.ksize = w0.data_ptr(), b0.data_ptr(), k0, kb
            self._keep += [w0, b0]
            for l, (gw, gb, wp, cb) in enumerate(layers):
                sb.gn_weight[l], sb.gn_bias[l] = gw.data_ptr(), gb.data_ptr()
                sb.conv_weight_packed[l], sb.conv_bias[l] = wp.data_ptr(), cb.data_ptr()
                self._keep += [gw, gb, wp, cb]
        a.image_stride = _strides4(image, (0, 1, 2, 3))
        a.feat_stride = _strides4(features, (0, 1, 2, 3))
        # shape / alignment query with placeholder pointers (nothing is dereferenced)
        a.image, a.features, a.out = image.data_ptr(), features.data_ptr(), image.data_ptr() & ~0xFF
        self.supported = lib.naf_forward_supported(C.byref(a)) == 1
        self.args, self.lib = a, lib
        self.out_dtype, self.shape_out = out_dtype, (B, Ho, Wo, Cc)
        self.ws_bytes = int(lib.naf_forward_workspace_bytes(C.byref(a))) if self.supported else 0
        # a forward that runs on ONE stream never touches the fourth activation buffer (C ABI 0.4.2: a quarter of the workspace at 1024^2)
        self.ws_bytes_one = int(lib.naf_forward_workspace_bytes_ex(C.byref(a), _lib.FWD_ONE_STREAM)) if self.supported else 0
        self._planned = {}
        self.key = (tuple(image.shape), tuple(image.stride()), image.dtype, tuple(features.shape), tuple(features.stride()), features.dtype,
                    (Ho, Wo))

    WS_POOL_BYTES = 8 << 30
    streams = 0            # 0: the library's plan (naf_forward_streams), 1: one stream, 2: two whenever possible
    conv0_exact = False    # NAF_FWD_CONV0_EXACT: exact fp32 products in the 3x3 first convolution (default: 16 mantissa bits)

    def planned_streams(self) -> int:
        """How many streams ``run`` will use (2 = the branches side by side on the lent stream)."""
        st = int(self.streams)
        if st not in self._planned:          # a pure function of the geometry and the flag: asked once
            flags = {0: 0, 1: _lib.FWD_ONE_STREAM, 2: _lib.FWD_TWO_STREAMS}[st]
            self._planned[st] = int(self.lib.naf_forward_streams(C.byref(self.args), flags))
        return self._planned[st]

    def release_workspaces(self, keep_stream: Optional[int] = None) -> None:
        """Drop the plan's scratch buffers (all of them, or all but the one of raw stream handle ``keep_stream``).  Safe once
        the work queued on those streams has been synchronised with; a captured graph keeps its own workspace alive itself."""
        pool = self.__dict__.get("_ws_by_stream", {})
        for k in [k for k in pool if keep_stream is None or k[1] != keep_stream]:
            del pool[k]
        if keep_stream is None:
            self._ws = None

    def view(self, which: str) -> torch.Tensor:
        """A tensor VIEW of an intermediate of the last ``run`` inside its workspace (naf_forward_workspace_view): ``"guidance"`` bf16
        [B, Ho, Wo, 256] (un-rotated), ``"keys"`` bf16 [B, h, w, 256], ``"values"`` bf16 [B, h, w, C].  Valid until the next run on
        that workspace; synchronise with the stream of the run before reading it on another one."""
        ws = self.__dict__.get("_ws")
        if ws is None:
            raise RuntimeError("ForwardPlan.view: no forward has run on this plan (or its workspace was released)")
        a = self.args
        code = {"guidance": 0, "keys": 1, "values": 2}[which]
        off, nbytes = C.c_size_t(0), C.c_size_t(0)
        _lib.check(self.lib.naf_forward_workspace_view(C.byref(a), code, C.byref(off), C.byref(nbytes)), "naf_forward_workspace_view")
        B, Ho, Wo, _ = self.shape_out
        shape = {"guidance": (B, Ho, Wo, 256), "keys": (B, a.h, a.w, 256), "values": (B, a.h, a.w, a.C)}[which]
        return ws[off.value: off.value + nbytes.value].view(torch.bfloat16).view(shape)

    def release_stream(self, device_index: int, stream_handle: int) -> None:
        """Drop the scratch buffer of ONE (device, raw stream handle) -- a throw-away stream's -- once its work has been synchronised with."""
        ws = self.__dict__.get("_ws_by_stream", {}).pop((device_index, stream_handle), None)
        if ws is not None and self.__dict__.get("_ws") is ws:
            self._ws = None

    def run(self, image: torch.Tensor, features: torch.Tensor, events=None, return_logits: bool = False, phase_events=None):
        a = self.args
        dev = image.device
        # the workspace belongs to the plan (one allocation per geometry, not per call) -- one per (device, stream) the plan has
        # been run on (round 3: forwards of one module on two streams no longer share scratch; the four most recent are kept).
        # Host-side state (the argument struct) is still per plan: calls from several THREADS need a module each.
        # Memory: a workspace is ~1 GB at 1024^2 and several GB at 2048^2, so the pool is bounded in BYTES too (WS_POOL_BYTES,
        # default 8 GiB: at least the current stream's workspace always stays); ``release_workspaces()`` drops all of them.
        skey = (dev.index, int(torch.cuda.current_stream(dev).cuda_stream))
        pool = self.__dict__.setdefault("_ws_by_stream", {})
        one = self.planned_streams() == 1       # then the call is made with NAF_FWD_ONE_STREAM and the smaller workspace is enough
        need = self.ws_bytes_one if one else self.ws_bytes
        ws = pool.pop(skey, None)
        if ws is not None and ws.numel() < need:
            ws = None
        if ws is None:
            while pool and (len(pool) >= 4 or (len(pool) + 1) * self.ws_bytes > self.WS_POOL_BYTES):
                pool.pop(next(iter(pool)))     # least recently used (dicts keep insertion order)
            ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        pool[skey] = ws
        self._ws = ws                           # the one the last call used (GraphedForward keeps it alive)
        out = torch.empty(self.shape_out, dtype=self.out_dtype, device=dev)
        a.image, a.features, a.out = image.data_ptr(), features.data_ptr(), out.data_ptr()
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.events[0] = events[0].cuda_event if events else None
        a.events[1] = events[1].cuda_event if events else None
        for i in range(8):     # naf_forward_args.phase_events: hipEvent_t handles at the phase boundaries of the one call
            e = phase_events[i] if (phase_events and i < len(phase_events)) else None    # None entries are skipped by the library
            a.phase_events[i] = e.cuda_event if e is not None else None
        logits = None
        if return_logits:
            logits = torch.empty((self.shape_out[0], a.heads, self.shape_out[1], self.shape_out[2], a.ksize * a.ksize),
                                 dtype=torch.float32, device=dev)
        a.logits = logits.data_ptr() if logits is not None else None
        flags = (_lib.FWD_CONV0_EXACT if self.conv0_exact else 0) | (_lib.FWD_ONE_STREAM if one else {0: 0, 2: _lib.FWD_TWO_STREAMS}[int(self.streams)])
        with torch.cuda.device(dev):
            if one:
                rc = self.lib.naf_forward_ex(C.byref(a), None, flags, _stream(image))
            else:
                with AUX_POOL.lease(skey[0], skey[1]) as aux:
                    rc = self.lib.naf_forward_ex(C.byref(a), C.byref(aux), flags, _stream(image))
        _lib.check(rc, "naf_forward_ex")
        out = out.permute(0, 3, 1, 2)
        return (out, logits) if return_logits else out
