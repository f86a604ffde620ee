"""ctypes binding of libnaf_hip.so (the C ABI declared in include/naf_hip.h).

The product path has NO fallback: if the shared library is missing or does not export every symbol,
``load()`` raises and every op built on it raises with it.  ``import torch`` happens first on purpose:
the library's DT_NEEDED libamdhip64.so.7 then resolves to the HIP runtime torch already mapped.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # noqa: F401  (must be imported before the HIP library is dlopen'ed)

# NAF_HIP_LIB lets an experiment point at an alternative build of the SAME library (A/B kernel variants)
LIB_PATH = os.environ.get("NAF_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libnaf_hip.so")

HEADER_VERSION = 402          # NAF_HIP_VERSION of the include/naf_hip.h these ctypes mirrors were written against
NAF_BF16, NAF_F32 = 0, 1
XNA_AUTO, XNA_MFMA, XNA_GENERIC, XNA_UNION, XNA_ROWS = 0, 1, 2, 3, 4

I64x4 = C.c_int64 * 4


class RopePoolArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("q", C.c_void_p), ("k_lr", C.c_void_p), ("tab_y", C.c_void_p), ("tab_x", C.c_void_p),
        ("x_dtype", C.c_int32), ("B", C.c_int32), ("Cq", C.c_int32), ("heads", C.c_int32), ("Ho", C.c_int32),
        ("Wo", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("x_stride", I64x4), ("q_stride", I64x4), ("k_stride", I64x4),
    ]


class XnaArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k_lr", C.c_void_p), ("v_lr", C.c_void_p), ("out", C.c_void_p), ("logits", C.c_void_p),
        ("idx_y", C.c_void_p), ("idx_x", C.c_void_p), ("rope_tab_y", C.c_void_p), ("rope_tab_x", C.c_void_p),
        ("B", C.c_int32), ("heads", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("h", C.c_int32),
        ("w", C.c_int32), ("Dq", C.c_int32), ("Dv", C.c_int32), ("ky", C.c_int32), ("kx", C.c_int32),
        ("out_dtype", C.c_int32), ("path", C.c_int32), ("scale", C.c_float), ("reserved", C.c_int32),
        ("q_stride", I64x4), ("k_stride", I64x4), ("v_stride", I64x4), ("o_stride", I64x4),
    ]


I64x3 = C.c_int64 * 3


class StemConv0Args(C.Structure):
    _fields_ = [
        ("image", C.c_void_p), ("y", C.c_void_p), ("weight", C.c_void_p), ("bias", C.c_void_p), ("stats_out", C.c_void_p),
        ("image_dtype", C.c_int32), ("ksize", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("channels", C.c_int32), ("image_stride", I64x4), ("y_stride", I64x3), ("flags", C.c_int32), ("reserved", C.c_int32),
    ]


CONV0_EXACT = 1                                  # naf_stem_conv0_args.flags
FWD_CONV0_EXACT, FWD_ONE_STREAM, FWD_TWO_STREAMS = 1, 2, 4   # naf_forward_ex flags


class StemConvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("gn_weight", C.c_void_p),
        ("gn_bias", C.c_void_p), ("stats_in", C.c_void_p), ("stats_out", C.c_void_p),
        ("ksize", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("eps", C.c_float),
        ("channels", C.c_int32), ("x_stride", I64x3), ("y_stride", I64x3), ("first", C.POINTER(StemConv0Args)),
    ]


class KeyPoolArgs(C.Structure):
    _fields_ = [
        ("k_lr", C.c_void_p), ("tab_y", C.c_void_p), ("tab_x", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32),
        ("k_stride", I64x3),
    ]


class RopePoolBwdArgs(C.Structure):
    _fields_ = [
        ("dq", C.c_void_p), ("dk_lr", C.c_void_p), ("dx", C.c_void_p), ("tab_y", C.c_void_p), ("tab_x", C.c_void_p),
        ("B", C.c_int32), ("Cq", C.c_int32), ("heads", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("h", C.c_int32),
        ("w", C.c_int32), ("dq_stride", I64x4), ("dk_stride", I64x4), ("dx_stride", I64x4),
    ]


class StemWgradArgs(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("gn_weight", C.c_void_p), ("gn_bias", C.c_void_p),
        ("stats_in", C.c_void_p), ("ksize", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("eps", C.c_float),
        ("channels", C.c_int32), ("dy_stride", I64x3), ("x_stride", I64x3),
    ]


class StemConv0WgradArgs(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("image", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("image_dtype", C.c_int32),
        ("ksize", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("channels", C.c_int32), ("dy_stride", I64x3),
        ("image_stride", I64x4),
    ]


class StemConv0DgradArgs(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("weight", C.c_void_p), ("dimage", C.c_void_p), ("ksize", C.c_int32), ("B", C.c_int32), ("H", C.c_int32),
        ("W", C.c_int32), ("channels", C.c_int32), ("accumulate", C.c_int32), ("dy_stride", I64x3), ("dimage_stride", I64x4),
    ]


class StemActArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("a", C.c_void_p), ("gn_weight", C.c_void_p), ("gn_bias", C.c_void_p), ("stats_in", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("channels", C.c_int32), ("pad", C.c_int32), ("eps", C.c_float),
        ("x_stride", I64x3), ("a_stride", I64x3),
    ]


class StemActBwdArgs(C.Structure):
    _fields_ = [
        ("da", C.c_void_p), ("x", C.c_void_p), ("dx", C.c_void_p), ("gn_weight", C.c_void_p), ("gn_bias", C.c_void_p),
        ("stats_in", C.c_void_p), ("sums", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("channels", C.c_int32), ("fold", C.c_int32), ("phase", C.c_int32),
        ("eps", C.c_float), ("da_stride", I64x3), ("x_stride", I64x3), ("dx_stride", I64x3),
    ]


class XnaBwdArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k_lr", C.c_void_p), ("v_lr", C.c_void_p), ("dout", C.c_void_p), ("dq", C.c_void_p),
        ("dk_lr", C.c_void_p), ("dv_lr", C.c_void_p), ("idx_y", C.c_void_p), ("idx_x", C.c_void_p),
        ("B", C.c_int32), ("heads", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("h", C.c_int32),
        ("w", C.c_int32), ("Dq", C.c_int32), ("Dv", C.c_int32), ("ky", C.c_int32), ("kx", C.c_int32),
        ("scale", C.c_float), ("path", C.c_int32),
        ("q_stride", I64x4), ("k_stride", I64x4), ("v_stride", I64x4), ("dout_stride", I64x4), ("dq_stride", I64x4),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
    ]


MAX_STEM_LAYERS = 8


class StemBranch(C.Structure):
    _fields_ = [
        ("conv0_weight", C.c_void_p), ("conv0_bias", C.c_void_p), ("conv0_ksize", C.c_int32), ("ksize", C.c_int32),
        ("gn_weight", C.c_void_p * MAX_STEM_LAYERS), ("gn_bias", C.c_void_p * MAX_STEM_LAYERS),
        ("conv_weight_packed", C.c_void_p * MAX_STEM_LAYERS), ("conv_bias", C.c_void_p * MAX_STEM_LAYERS),
    ]


class ForwardArgs(C.Structure):
    _fields_ = [
        ("image", C.c_void_p), ("features", C.c_void_p), ("out", C.c_void_p), ("tab_y", C.c_void_p), ("tab_x", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("events", C.c_void_p * 2), ("logits", C.c_void_p),
        ("branch", StemBranch * 2),
        ("nlayer", C.c_int32), ("image_dtype", C.c_int32), ("feat_dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("C", C.c_int32),
        ("heads", C.c_int32), ("ksize", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("heads_rope", C.c_int32), ("reserved", C.c_int32), ("gn_eps", C.c_float), ("scale", C.c_float),
        ("image_stride", I64x4), ("feat_stride", I64x4), ("phase_events", C.c_void_p * 8),
    ]


class ForwardAux(C.Structure):
    """naf_forward_aux: the second stream and the fork / join events the caller lends to naf_forward_ex."""
    _fields_ = [("stream", C.c_void_p), ("fork_event", C.c_void_p), ("join_event", C.c_void_p)]


# Entry points that read or write GroupNorm-sum buffers are EXPORTED under names that carry the copy count (naf_hip.h, 0.4.0: a
# binary built for another buffer layout fails to resolve them instead of overrunning its buffers); header name -> exported name
STATS_SLOTS_ABI = 16
EXPORTED_AS = {n: f"{n}_s{STATS_SLOTS_ABI}" for n in ("naf_stem_conv0_fwd", "naf_stem_conv_fwd", "naf_stem_conv_keys_fwd",
                                                      "naf_stem_act_fwd", "naf_stem_act_bwd", "naf_stem_wgrad")}

# symbol -> (restype, argtypes); must list every function include/naf_hip.h declares
SIGNATURES = {
    "naf_version": (C.c_int, []),
    "naf_last_error": (C.c_char_p, []),
    "naf_abi_check": (C.c_int, [C.c_int]),
    "naf_stem_stats_bytes": (C.c_size_t, [C.c_int32]),
    "naf_forward_aux_create": (C.c_int, [C.POINTER(ForwardAux)]),
    "naf_forward_aux_destroy": (C.c_int, [C.POINTER(ForwardAux)]),
    "naf_forward_streams": (C.c_int, [C.POINTER(ForwardArgs), C.c_uint32]),
    "naf_forward_workspace_view": (C.c_int, [C.POINTER(ForwardArgs), C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "naf_forward_ex": (C.c_int, [C.POINTER(ForwardArgs), C.POINTER(ForwardAux), C.c_uint32, C.c_void_p]),
    "naf_axis_index_table": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32]),
    "naf_stem_weight_index": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "naf_axis_index_table_device": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "naf_stem_conv0_fwd": (C.c_int, [C.POINTER(StemConv0Args), C.c_void_p]),
    "naf_stem_conv_fwd": (C.c_int, [C.POINTER(StemConvArgs), C.c_void_p]),
    "naf_stem_conv_keys_supported": (C.c_int, [C.POINTER(StemConvArgs), C.POINTER(KeyPoolArgs)]),
    "naf_stem_conv_keys_fwd": (C.c_int, [C.POINTER(StemConvArgs), C.POINTER(KeyPoolArgs), C.c_void_p]),
    "naf_rope_pool_bwd": (C.c_int, [C.POINTER(RopePoolBwdArgs), C.c_void_p]),
    "naf_stem_wgrad": (C.c_int, [C.POINTER(StemWgradArgs), C.c_void_p]),
    "naf_stem_conv0_wgrad": (C.c_int, [C.POINTER(StemConv0WgradArgs), C.c_void_p]),
    "naf_stem_conv0_dgrad": (C.c_int, [C.POINTER(StemConv0DgradArgs), C.c_void_p]),
    "naf_stem_act_fwd": (C.c_int, [C.POINTER(StemActArgs), C.c_void_p]),
    "naf_stem_act_bwd": (C.c_int, [C.POINTER(StemActBwdArgs), C.c_void_p]),
    "naf_rope_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "naf_rope_pool_fwd": (C.c_int, [C.POINTER(RopePoolArgs), C.c_void_p]),
    "naf_preshrink_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(I64x4), C.c_void_p]),
    "naf_pool_guidance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "naf_pack_values": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(C.c_int64), C.c_void_p]),
    "naf_xna_select": (C.c_int, [C.POINTER(XnaArgs)]),
    "naf_xna_union_plan": (C.c_int, [C.POINTER(XnaArgs), C.POINTER(C.c_int32)]),
    "naf_workspace_bytes": (C.c_size_t, [C.POINTER(XnaArgs)]),
    "naf_xna_fwd": (C.c_int, [C.POINTER(XnaArgs), C.c_void_p]),
    "naf_xna_bwd_supported": (C.c_int, [C.POINTER(XnaBwdArgs)]),
    "naf_xna_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(XnaBwdArgs)]),
    "naf_xna_bwd_chunk_plan": (C.c_int, [C.POINTER(XnaBwdArgs), C.POINTER(C.c_int32), C.c_int]),
    "naf_xna_bwd": (C.c_int, [C.POINTER(XnaBwdArgs), C.c_void_p]),
    "naf_forward_workspace_bytes": (C.c_size_t, [C.POINTER(ForwardArgs)]),
    "naf_forward_workspace_bytes_ex": (C.c_size_t, [C.POINTER(ForwardArgs), C.c_uint32]),
    "naf_forward_supported": (C.c_int, [C.POINTER(ForwardArgs)]),
    "naf_forward": (C.c_int, [C.POINTER(ForwardArgs), C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


class NafHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libnaf_hip.so and bind every declared symbol; raises NafHipError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NafHipError(
                f"{LIB_PATH} not found: build it with `python -m naf_amd.build` (hipcc, gfx950). "
                "naf_amd has no CPU or PyTorch fallback for its kernels.")
        try:
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        except OSError as e:  # pragma: no cover
            raise NafHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            exported = EXPORTED_AS.get(name, name)
            try:
                fn = getattr(lib, exported)
            except AttributeError as e:
                raise NafHipError(f"{LIB_PATH} does not export {exported}; rebuild with `python -m naf_amd.build --force`") from e
            fn.restype, fn.argtypes = res, args
            if exported != name:
                setattr(lib, name, fn)           # callers use the header's names
        rc = lib.naf_abi_check(HEADER_VERSION)
        if rc != 0:
            raise NafHipError(f"{LIB_PATH}: {lib.naf_last_error().decode('utf-8', 'replace')}")
        _lib = lib
    return _lib


def last_error() -> str:
    return load().naf_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = last_error()
        if rc == 1:
            raise ValueError(f"{what}: {msg}")
        raise NafHipError(f"{what} failed (status {rc}): {msg}")
