"""CPU tests of the drop-in boundary: libnaf_hip.so builds for gfx950, loads, exports every symbol that
include/naf_hip.h declares; the host-side functions and the Python mirror of the reference interface
behave like the reference (names, arguments, errors).  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import naf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_text():
    txt = open(os.path.join(ROOT, "include", "naf_hip.h")).read()
    return re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


def _declared_symbols():
    """Function names as the header's prototypes spell them."""
    return sorted(set(re.findall(r"\b(naf_[a-z_0-9]+)\s*\(", _header_text())))


def _export_names():
    """header name -> exported name: the `#define naf_x NAF_ABI_PASTE(naf_x, NAF_STATS_SLOTS)` lines that version the entry points by
    buffer layout (0.4.0; since round 6 the suffix is pasted from NAF_STATS_SLOTS instead of spelled out)."""
    txt = _header_text()
    slots = int(re.search(r"#define\s+NAF_STATS_SLOTS\s+(\d+)", txt).group(1))
    names = re.findall(r"^#define\s+(naf_[a-z_0-9]+)\s+NAF_ABI_PASTE\((naf_[a-z_0-9]+),\s*NAF_STATS_SLOTS\)\s*$", txt, flags=re.M)
    assert all(a == b for a, b in names)
    return {a: f"{a}_s{slots}" for a, _ in names}


def test_library_builds_and_exports_every_declared_symbol(built_lib):
    lib = C.CDLL(built_lib)
    declared = _declared_symbols()
    assert {"naf_version", "naf_last_error", "naf_axis_index_table", "naf_rope_tables", "naf_rope_pool_fwd",
            "naf_stem_conv0_fwd", "naf_stem_conv_fwd",
            "naf_pack_values", "naf_xna_select", "naf_workspace_bytes", "naf_xna_fwd"} <= set(declared)
    exported = _export_names()
    for name in declared:
        assert hasattr(lib, exported.get(name, name)), f"libnaf_hip.so does not export {exported.get(name, name)}"
    from naf_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared, "ctypes binding and header disagree"
    assert exported == _lib.EXPORTED_AS, "ctypes binding and header disagree on the versioned export names"
    lib.naf_version.restype = C.c_int
    assert lib.naf_version() >= 100


def test_stale_binaries_fail_loudly(built_lib):
    """C ABI 0.4.0: (1) every entry point that reads or writes GroupNorm-sum buffers is exported ONLY under a name that carries the
    copy count, so a binary built against a 0.2.x / 0.3.x header (one copy / unversioned names) fails to resolve it instead of
    overrunning its buffers; (2) naf_abi_check tells a host compiled against another minor version so; (3) naf_stem_stats_bytes
    is the size of one buffer as this library lays it out."""
    from naf_amd import _lib, ops
    lib = C.CDLL(built_lib)
    exported = _export_names()
    assert set(exported) == {"naf_stem_conv0_fwd", "naf_stem_conv_fwd", "naf_stem_conv_keys_fwd", "naf_stem_act_fwd", "naf_stem_act_bwd", "naf_stem_wgrad"}
    for old, new in exported.items():
        assert new == f"{old}_s{ops.STATS_SLOTS}"
        assert hasattr(lib, new)
        with pytest.raises(AttributeError):
            getattr(lib, old)                    # the name an old binary asks for is gone
    lib.naf_abi_check.restype, lib.naf_abi_check.argtypes = C.c_int, [C.c_int]
    lib.naf_last_error.restype = C.c_char_p
    lib.naf_version.restype = C.c_int
    v = lib.naf_version()
    assert v == _lib.HEADER_VERSION
    assert lib.naf_abi_check(v) == 0 and lib.naf_abi_check(v - v % 100) == 0 and lib.naf_abi_check(v - v % 100 + 99) == 0     # patch level is free
    for stale in (300, 205, v + 100):
        assert lib.naf_abi_check(stale) == 1
        msg = lib.naf_last_error().decode()
        assert f"{stale // 10000}.{stale // 100 % 100}.{stale % 100}" in msg and "rebuild" in msg
    lib.naf_stem_stats_bytes.restype, lib.naf_stem_stats_bytes.argtypes = C.c_size_t, [C.c_int32]
    assert lib.naf_stem_stats_bytes(3) == ops.new_stats(3, "cpu").numel() * 8 == 16 * 3 * 16 * 8
    assert lib.naf_stem_stats_bytes(0) == 0 and lib.naf_stem_stats_bytes(-2) == 0


def test_forward_aux_contract_without_a_device(built_lib):
    """naf_forward_aux / naf_forward_ex argument checks that need no GPU: the library owns no stream (0.4.0), a NULL aux is the
    one-stream forward, destroying an all-NULL bundle is a no-op, contradictory flags and half-filled bundles are refused."""
    from naf_amd import _lib
    lib = _lib.load()
    aux = _lib.ForwardAux()
    assert lib.naf_forward_aux_destroy(C.byref(aux)) == 0
    assert lib.naf_forward_aux_destroy(None) == 1 and "NULL" in _lib.last_error()
    assert lib.naf_forward_aux_create(None) == 1
    assert C.sizeof(_lib.ForwardAux) == 3 * C.sizeof(C.c_void_p)
    assert lib.naf_forward_ex(None, None, 0, None) == 1                      # NULL args: invalid, like naf_forward
    assert lib.naf_forward(None, None) == 1


def test_groupnorm_sum_copies_match_header_and_version(built_lib):
    """The number of partial copies of the GroupNorm-sum buffers is part of the ABI (0.3.0): header, Python host and the
    version the library reports agree; the host-side helpers keep the [copies, B, 8, 2] layout."""
    from naf_amd import ops
    txt = open(os.path.join(ROOT, "include", "naf_hip.h")).read()
    slots = int(re.search(r"#define\s+NAF_STATS_SLOTS\s+(\d+)", txt).group(1))
    version = int(re.search(r"#define\s+NAF_HIP_VERSION\s+(\d+)", txt).group(1))
    assert slots == ops.STATS_SLOTS == 16 and version >= 300
    lib = C.CDLL(built_lib)
    lib.naf_version.restype = C.c_int
    assert lib.naf_version() == version
    st = ops.new_stats(3, "cpu", lead=(2,))
    assert tuple(st.shape) == (2, slots, 3, 8, 2) and st.dtype == torch.float64 and float(st.abs().sum()) == 0.0
    tot = torch.arange(3 * 8 * 2, dtype=torch.float64).view(3, 8, 2)
    buf = ops.stats_from_total(tot)
    assert tuple(buf.shape) == (slots, 3, 8, 2) and torch.equal(ops.stats_total(buf), tot) and float(buf[1:].abs().sum()) == 0.0
    with pytest.raises(ValueError, match="GroupNorm sums"):
        ops._stats_ptr(tot, 3, "test")          # a 0.2.x-shaped buffer is refused before any launch


@pytest.mark.parametrize("ks,Cc", [(3, 128), (1, 128), (3, 64), (1, 48), (3, 256)])
def test_packed_weight_order_matches_the_library(built_lib, ks, Cc):
    """ops.pack_conv_weight puts element (tap, oc, ic) where naf_stem_weight_index says the kernels read it (register order for
    the 3x3 layers of the default width, [tap][oc][ic] everywhere else), unpack_conv_weight inverts it, and the index is a
    permutation of the k*k*C*C elements; out-of-range arguments give -1."""
    from naf_amd import ops
    lib = C.CDLL(built_lib)
    f = lib.naf_stem_weight_index
    f.restype = C.c_int64
    f.argtypes = [C.c_int32] * 5
    w = torch.arange(Cc * Cc * ks * ks, dtype=torch.float32).view(Cc, Cc, ks, ks) % 251          # exact in bf16
    wp = ops.pack_conv_weight(w)
    assert tuple(wp.shape) == (ks * ks, Cc, Cc) and wp.dtype == torch.bfloat16 and wp.is_contiguous()
    flat = wp.float().flatten()
    rng = np.random.RandomState(ks * 1000 + Cc)
    seen = set()
    for _ in range(3000):
        t, oc, ic = int(rng.randint(ks * ks)), int(rng.randint(Cc)), int(rng.randint(Cc))
        idx = f(ks, Cc, t, oc, ic)
        assert 0 <= idx < flat.numel()
        assert float(flat[idx]) == float(w[oc, ic, t // ks, t % ks]), (t, oc, ic, idx)
        seen.add(idx)
    if ks == 3 and Cc == 128:
        all_idx = {f(3, 128, t, oc, ic) for t in range(9) for oc in range(0, 128, 7) for ic in range(128)}
        assert len(all_idx) == 9 * len(range(0, 128, 7)) * 128                                   # injective
        assert f(3, 128, 0, 1, 0) == 8 and f(3, 128, 0, 0, 8) == 32 * 8 and f(3, 128, 0, 0, 16) == 64 * 8 and f(3, 128, 0, 32, 0) == 8 * 64 * 8
        assert f(3, 0, 4, 5, 6) == f(3, 128, 4, 5, 6)                                            # channels 0 = the default width
    else:
        assert f(ks, Cc, 0, 1, 0) == Cc and f(ks, Cc, 0, 0, 1) == 1
    assert torch.equal(ops.unpack_conv_weight(wp), w)
    assert f(ks, Cc, ks * ks, 0, 0) == -1 and f(ks, Cc, 0, Cc, 0) == -1 and f(2, Cc, 0, 0, 0) == -1 and f(ks, 24, 0, 0, 0) == -1


def test_struct_layout_matches_header(built_lib):
    """sizeof of the ctypes mirrors == the C structs (checked through a tiny C probe compiled with gcc)."""
    import subprocess, tempfile
    from naf_amd import _lib
    src = ('#include "naf_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(naf_rope_pool_args), '
           'sizeof(naf_xna_args), sizeof(naf_stem_conv0_args), sizeof(naf_stem_conv_args), sizeof(naf_xna_bwd_args), '
           'sizeof(naf_forward_args));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        a, b, c, e, f, g = map(int, subprocess.check_output([os.path.join(d, "p")]).split())
    assert a == C.sizeof(_lib.RopePoolArgs) and b == C.sizeof(_lib.XnaArgs)
    assert c == C.sizeof(_lib.StemConv0Args) and e == C.sizeof(_lib.StemConvArgs)
    assert f == C.sizeof(_lib.XnaBwdArgs) and g == C.sizeof(_lib.ForwardArgs)
    # phase_events was appended (0.1.3): the last 8 pointers of the struct, older fields where they were
    assert _lib.ForwardArgs.phase_events.offset == g - 8 * C.sizeof(C.c_void_p)
    assert _lib.ForwardArgs.feat_stride.offset == g - 8 * C.sizeof(C.c_void_p) - 32


def test_training_struct_layouts_match_header(built_lib):
    """Size AND the offset of the last field of every training-side argument struct (round 2) against gcc's view of the header."""
    import subprocess, tempfile
    from naf_amd import _lib
    pairs = [("naf_stem_act_args", _lib.StemActArgs, "a_stride"), ("naf_stem_act_bwd_args", _lib.StemActBwdArgs, "dx_stride"),
             ("naf_stem_wgrad_args", _lib.StemWgradArgs, "x_stride"), ("naf_stem_conv0_wgrad_args", _lib.StemConv0WgradArgs, "image_stride"),
             ("naf_rope_pool_bwd_args", _lib.RopePoolBwdArgs, "dx_stride"), ("naf_xna_bwd_args", _lib.XnaBwdArgs, "workspace_bytes")]
    body = "".join(f'printf("%zu %zu\\n", sizeof({c}), offsetof({c}, {last}));' for c, _, last in pairs)
    src = '#include "naf_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){' + body + 'return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        rows = [tuple(map(int, l.split())) for l in subprocess.check_output([os.path.join(d, "p")]).decode().splitlines()]
    for (cname, ct, last), (size, off) in zip(pairs, rows):
        assert size == C.sizeof(ct), cname
        assert off == getattr(ct, last).offset, (cname, last)


def test_hip_stem_parameter_order():
    """_HipStem.backward returns its gradients in the order _hip_stem_params lists the parameters: per branch conv0 (w, b),
    then per layer (norm.w, norm.b, conv.w, conv.b) -- all 36 encoder parameters, each once."""
    from naf_amd.model import NAF, _hip_stem_params, _stem_layers
    m = NAF()
    ps = _hip_stem_params(m.image_encoder)
    enc_params = [p for n, p in m.named_parameters() if n.startswith("image_encoder.") and p.requires_grad]
    assert len(ps) == 36 and {id(p) for p in ps} == {id(p) for p in enc_params}
    assert ps[0] is m.image_encoder.encoder[0].weight and ps[1] is m.image_encoder.encoder[0].bias
    n0, c0 = _stem_layers(m.image_encoder.encoder)[0]
    assert ps[2] is n0.weight and ps[3] is n0.bias and ps[4] is c0.weight and ps[5] is c0.bias
    assert ps[18] is m.image_encoder.sem_encoder[0].weight


@pytest.mark.parametrize("L_in", [1, 2, 3, 5, 7, 14, 28, 32, 64])
def test_axis_index_table_matches_oracle(built_lib, L_in):
    from naf_amd import ops
    n = 0
    for L_out in list(range(L_in, 4 * L_in + 3)) + [16 * L_in, 16 * L_in + 5, 32 * L_in]:
        for k in (1, 3, 5, 7, 9, 15):
            if k * (L_out // L_in) > L_out:
                with pytest.raises(ValueError):
                    ops.axis_index_table(L_out, L_in, k)
                continue
            t = ops.axis_index_table(L_out, L_in, k).numpy()
            assert np.array_equal(t, O.axis_index_table(L_out, L_in, k)), (L_out, L_in, k)
            assert t.min() >= 0 and t.max() < L_in
            n += 1
    assert n > 0


def test_axis_index_table_errors_mirror_natten(built_lib):
    from naf_amd import ops
    with pytest.raises(ValueError, match="odd"):
        ops.axis_index_table(16, 4, 4)
    with pytest.raises(ValueError, match="exceeds"):
        ops.axis_index_table(20, 5, 7)
    with pytest.raises(ValueError, match="smaller"):
        ops.axis_index_table(4, 8, 3)


def _xna_args(h, w, Ho, Wo, Cc, k, heads=4, B=1, Dq=64, path=0, logits=False):
    from naf_amd._lib import XnaArgs, I64x4
    a = XnaArgs()
    Dv = Cc // heads
    a.q = a.k_lr = a.v_lr = a.out = 0x1000           # host logic only: never dereferenced
    a.logits = 0x1000 if logits else None
    a.B, a.heads, a.Ho, a.Wo, a.h, a.w, a.Dq, a.Dv, a.ky, a.kx = B, heads, Ho, Wo, h, w, Dq, Dv, k, k
    a.out_dtype, a.path, a.scale = 0, path, 0.0
    a.q_stride = I64x4(Ho * Wo * heads * Dq, Dq, Wo * heads * Dq, heads * Dq)
    a.k_stride = I64x4(h * w * heads * Dq, Dq, w * heads * Dq, heads * Dq)
    a.v_stride = I64x4(h * w * heads * Dv, Dv, w * heads * Dv, heads * Dv)
    a.o_stride = I64x4(Ho * Wo * heads * Dv, Dv, Wo * heads * Dv, heads * Dv)
    return a


@pytest.mark.parametrize("geom", [(28, 28, 64, 64, 384, 9), (37, 37, 512, 512, 768, 9), (5, 7, 23, 30, 64, 3),
                                  (64, 64, 256, 256, 768, 7), (128, 128, 256, 256, 768, 7), (48, 40, 48, 40, 384, 7),
                                  (33, 47, 33, 47, 64, 15), (21, 21, 50, 50, 2048, 13), (20, 17, 300, 40, 64, 11)])
def test_union_planner_covers_every_window(built_lib, geom):
    """Host logic of the table-driven MFMA path: the planned LDS rectangle must contain the taps of every query of
    every workgroup, a tile's taps must fit its slots, and the plan must fit the 160 KB LDS."""
    from naf_amd import _lib, ops
    h, w, Ho, Wo, Cc, k = geom
    lib = _lib.load()
    a = _xna_args(*geom, path=_lib.XNA_UNION)
    out = (C.c_int32 * 7)()
    assert lib.naf_xna_union_plan(C.byref(a), out) == 1
    wt, ry, seg, hub, wub, dvt, lds = list(out)
    assert wt in (16, 32) and seg % 16 == 0 and ry >= 1 and (Cc // 4) % dvt == 0 and dvt % 16 == 0 and lds <= 160 * 1024
    ty, tx = ops.axis_index_table(Ho, h, k).numpy(), ops.axis_index_table(Wo, w, k).numpy()

    def span(t, blk):
        return max(int(t[i:i + blk, -1].max() - t[i:i + blk, 0].min() + 1) for i in range(0, t.shape[0], blk))

    assert span(tx, 16) <= wt
    assert span(ty, ry) == hub and span(tx, seg) == wub
    assert lds >= (hub * wub + 32) * (72 + dvt + 16) * 2
    assert lib.naf_xna_select(C.byref(a)) == _lib.XNA_UNION


def test_xna_auto_policy(built_lib):
    """AUTO: cell kernels for integer ratios with 10x10+ cells, the table-driven MFMA kernel for smaller cells,
    non-integer ratios and ratio 1, the row-streaming MFMA kernel for other head dims / few value channels, the generic
    kernel for what is left (odd head dims, return_weights off the cell path)."""
    from naf_amd import _lib
    lib = _lib.load()
    sel = lambda *g, **kw: lib.naf_xna_select(C.byref(_xna_args(*g, **kw)))
    assert sel(64, 64, 1024, 1024, 768, 7) == _lib.XNA_MFMA
    assert sel(37, 37, 518, 518, 768, 9) == _lib.XNA_MFMA            # 14x14 cells
    assert sel(64, 64, 512, 512, 768, 7) == _lib.XNA_UNION           # 8x8 cells
    assert sel(28, 28, 64, 64, 384, 9) == _lib.XNA_UNION             # ratio 2.29
    assert sel(40, 40, 40, 40, 384, 7) == _lib.XNA_UNION             # ratio 1
    assert sel(28, 28, 64, 64, 384, 9, logits=True) == _lib.XNA_GENERIC
    assert sel(24, 20, 24, 20, 3, 5, heads=1, Dq=96) == _lib.XNA_ROWS        # denoising-like: one head of 96, C = 3
    assert sel(24, 20, 24, 20, 3, 5, heads=1, Dq=80) == _lib.XNA_GENERIC     # head dim without an instantiation
    assert sel(24, 20, 24, 20, 3, 5, heads=1, Dq=96, logits=True) == _lib.XNA_GENERIC
    assert sel(8, 8, 32, 32, 128, 7, logits=True) == _lib.XNA_MFMA   # return_weights stays on the cell kernel
    assert sel(28, 28, 64, 64, 384, 9, path=_lib.XNA_MFMA) == -2     # NAF_ERR_UNSUPPORTED


def _xna_bwd_args(h, w, Ho, Wo, Cc, k, heads=4, B=1, Dq=64, path=0):
    from naf_amd._lib import XnaBwdArgs, I64x4
    a = XnaBwdArgs()
    Dv = Cc // heads
    a.q = a.k_lr = a.v_lr = a.dout = a.dq = a.dk_lr = a.dv_lr = 0x1000      # host logic only: never dereferenced
    a.B, a.heads, a.Ho, a.Wo, a.h, a.w, a.Dq, a.Dv, a.ky, a.kx = B, heads, Ho, Wo, h, w, Dq, Dv, k, k
    a.scale, a.path = 0.0, path
    a.q_stride = a.dq_stride = I64x4(Ho * Wo * heads * Dq, Dq, Wo * heads * Dq, heads * Dq)
    a.k_stride = I64x4(h * w * heads * Dq, Dq, w * heads * Dq, heads * Dq)
    a.v_stride = I64x4(h * w * heads * Dv, Dv, w * heads * Dv, heads * Dv)
    a.dout_stride = I64x4(Ho * Wo * heads * Dv, Dv, Wo * heads * Dv, heads * Dv)
    return a


def test_xna_backward_policy_and_chunk_plan(built_lib):
    """Host logic of naf_xna_bwd (no device call): which kernel AUTO picks, what naf_xna_bwd_args.path (0.4.1) insists on, and the channel
    chunks the cell kernels run wide heads in at the large windows -- every chunk a width the kernels are instantiated for, the chunks adding
    up to Dv, one launch up to 9 x 9 (/root/reference/train.py:127-137 is what the call replaces)."""
    from naf_amd import _lib
    lib = _lib.load()
    sel = lambda *g, **kw: lib.naf_xna_bwd_supported(C.byref(_xna_bwd_args(*g, **kw)))

    def plan(*g, **kw):
        out = (C.c_int32 * 8)()
        n = lib.naf_xna_bwd_chunk_plan(C.byref(_xna_bwd_args(*g, **kw)), out, 8)
        return list(out[:n]) if n >= 0 else n

    assert sel(64, 64, 1024, 1024, 768, 7) == _lib.XNA_MFMA                  # G1
    assert sel(32, 32, 512, 512, 1024, 15) == _lib.XNA_MFMA                  # G2's largest window (round 5: channel chunks)
    assert sel(16, 16, 32, 32, 768, 9) == _lib.XNA_ROWS                      # the reference's own training geometry (ratio 2)
    assert sel(28, 28, 392, 392, 384, 9) == _lib.XNA_MFMA                    # patch-14 backbone: 14 of a row tile's 16 lanes (0.4.2: partial row tiles, windows <= 9)
    assert sel(28, 28, 392, 392, 384, 11) == _lib.XNA_ROWS                   # ... an 11 x 11 window at that ratio stays on the row-streaming kernel
    assert sel(28, 28, 308, 308, 384, 9) == _lib.XNA_ROWS                    # ratio 11: 5 of 16 lanes would idle (xna_row_tiles_ok), no row tiles
    assert sel(24, 20, 24, 20, 3, 5, heads=1, Dq=80) == _lib.XNA_GENERIC     # head dim without a matrix-core instantiation
    assert sel(64, 64, 1024, 1024, 768, 7, path=_lib.XNA_GENERIC) == _lib.XNA_GENERIC   # the tests' independent reference, any shape
    assert sel(64, 64, 1024, 1024, 768, 7, path=_lib.XNA_ROWS) == _lib.XNA_ROWS
    assert sel(16, 16, 32, 32, 768, 9, path=_lib.XNA_MFMA) == -2             # NAF_ERR_UNSUPPORTED: insisting on a kernel that does not apply
    assert sel(64, 64, 1024, 1024, 768, 7, path=3) == -1                     # NAF_XNA_UNION is a forward path: NAF_ERR_INVALID
    for k in (3, 5, 7, 9):
        for C_ in (128, 256, 384, 512, 768, 1024):
            assert plan(2 * k, 2 * k, 32 * k, 32 * k, C_, k) == [C_ // 4]    # the whole head in one launch
    widths = {11: 128, 13: 64, 15: 64}
    for k, lim in widths.items():
        for C_ in (128, 256, 384, 512, 768, 1024):
            Dv = C_ // 4
            got = plan(2 * k, 2 * k, 32 * k, 32 * k, C_, k)
            assert sum(got) == Dv and all(c in (32, 64, 96, 128) and c <= lim for c in got), (k, Dv, got)
            assert len(got) == -(-Dv // lim), (k, Dv, got)                   # as few launches as the chunk limit allows
    assert plan(32, 32, 512, 512, 1024, 15) == [64, 64, 64, 64] and plan(32, 32, 512, 512, 1024, 11) == [128, 128]
    assert plan(32, 32, 512, 512, 768, 11) == [96, 96] and plan(32, 32, 512, 512, 384, 15) == [64, 32]
    assert plan(16, 16, 32, 32, 768, 9) == []                                # another kernel serves the call
    assert lib.naf_xna_bwd_chunk_plan(C.byref(_xna_bwd_args(32, 32, 512, 512, 1024, 15)), None, 0) == 4
    assert lib.naf_xna_bwd_chunk_plan(C.byref(_xna_bwd_args(32, 32, 512, 512, 1024, 15)), None, 4) == -1


def test_module_mirrors_reference_interface():
    from naf_amd import NAF
    m = NAF()
    sd = m.state_dict()
    ref_keys = set(O.make_params().keys())
    assert set(sd.keys()) == ref_keys                       # 36 conv/GN tensors + rope periods
    assert sum(p.numel() for p in m.parameters()) == 662528   # test/test_results.json:255 "# Params"
    assert m.upsampler.kernel_size == (9, 9) and m.upsampler.num_heads == 4
    assert abs(m.upsampler.scale - 0.125) < 1e-12
    assert torch.equal(m.image_encoder.rope.periods, O.rope_periods(256, 4, 100.0))
    m.load_state_dict(O.make_params(seed=3), strict=True)
    m2 = NAF(dim=96, heads_attn=1, heads_rope=1, kernel_size=15, img_layers=2, use_semencoder=True)   # denoising.py notes
    assert m2.upsampler.kernel_size == (15, 15)
    with pytest.raises(AssertionError):
        NAF(dim=128, heads_attn=3)          # attentions.py:41 "dim must be divisible by num_heads"


def test_forward_has_no_cpu_fallback():
    from naf_amd import NAF
    m = NAF()
    with pytest.raises(RuntimeError, match="ROCm"):
        m(torch.zeros(1, 3, 16, 16), torch.zeros(1, 8, 4, 4), (16, 16))


def test_product_never_imports_oracle():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import naf_amd, naf_amd.ops, naf_amd.dist, hubconf; "
            "bad=[m for m in sys.modules if m.split('.')[0]=='oracle']; assert not bad, bad") % ROOT
    subprocess.check_call([sys.executable, "-c", code])
    for dirpath, _, files in os.walk(os.path.join(ROOT, "naf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("no oracle", ""), f


def test_hubconf_entry_point():
    import hubconf
    m = hubconf.naf(pretrained=False, device="cpu")
    assert m.training and type(m).__name__ == "NAF"          # the reference does not call .eval() either (hubconf.py:20-24)
    assert "naf" in dir(hubconf) and hubconf.dependencies == ["torch"]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from naf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NafHipError, match="no CPU or PyTorch fallback"):
        _lib.load()
