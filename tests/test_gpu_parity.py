"""GPU parity tests (-m gpu): every HIP kernel, called through the C ABI, against the CPU oracle on the
same seeded inputs and against the golden vectors produced by the imported reference.

Tolerances (fp, stated here as the prompt asks):
  * kernels fed bf16-rounded inputs vs the fp32 oracle on the SAME rounded inputs:
      fp32 output  : |err| <= 6e-3 + 6e-3*|ref|   (P is rounded to bf16 before the PV MFMA: rel 2^-9 per weight)
      bf16 output  : |err| <= 1.2e-2 + 1.2e-2*|ref|  (+ one bf16 rounding of the result)
  * whole forward (bf16-activation HIP conv stem, bf16 Q/K/V) vs the fp32 reference golden vectors / the oracle:
      SURVEY.md section 8c's |err| <= 2e-2 + 1e-2*|ref| elementwise and mean |err| <= 6e-3 (outputs are O(1));
      looser only where the softmax is peaked and the output follows single keys instead of averaging them (the stem's bf16
      error, 6.9e-3 of the guidance RMS after five layers -- profiles/r02_stem_error_budget.txt -- then shows undamped), each
      with its measured budget (profiles/r04_tolerance_budget.txt, tools/tolerance_probe.py: measured maximum x 1.5 and a bound
      on the fraction of elements outside 8c): cells of 2 pixels 3.6e-2, of 1 pixel 6e-2 (+ 1e-2*|ref|), the one-head k=5
      denoising golden F6 4.5e-2 (profiles/r03_f6_error.txt).  Everything else, bf16 I/O and 14-pixel cells included, holds 8c.
"""
import os

import numpy as np
import pytest
import torch

from oracle import naf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    from naf_amd import _lib
    _lib.load()                      # fail loudly if the HIP library is missing on a GPU box
    return torch.device("cuda:0")


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def oracle_xna_backward(dev, q, k, v, dout, ksz, heads):
    """``O.xna_backward`` (fp64 autograd through the oracle's forward).  Cases whose gathered windows are large -- wide heads with
    9 x 9 ... 15 x 15 windows: 10-25 s of host time each, a third of the GPU suite in round 5 -- are evaluated by the SAME oracle
    code on the device (ATen's fp64 kernels; nothing hand-written); test_oracle_on_the_device_equals_the_oracle_on_the_host holds
    that evaluation to the host's on this very box."""
    B, _, Ho, Wo = q.shape
    kk = ksz * ksz if isinstance(ksz, int) else ksz[0] * ksz[1]
    work = B * Ho * Wo * kk * (q.shape[1] + v.shape[1])
    if work < 1.5e8:
        return O.xna_backward(q, k, v, dout, ksz, heads)
    return tuple(t.cpu() for t in O.xna_backward(q.to(dev), k.to(dev), v.to(dev), dout.to(dev), ksz, heads))


def to5(x, heads):
    """[B, C, H, W] fp32 -> bf16 5-D [B, heads, H, W, D] (head-major, D contiguous)."""
    B, C, H, W = x.shape
    return x.view(B, heads, C // heads, H, W).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)


def run_xna(dev, q, k, v, ksz, heads, out_dtype=torch.float32, path="auto", return_logits=False):
    from naf_amd import ops
    B, C = v.shape[:2]
    h, w = v.shape[-2:]
    q5, k5 = to5(q, heads).to(dev), to5(k, heads).to(dev)
    vp = ops.pack_values(v.to(dev))
    v5 = vp.view(B, h, w, heads, C // heads).permute(0, 3, 1, 2, 4)
    res = ops.xna_forward(q5, k5, v5, ksz, out_dtype=out_dtype, path=path, return_logits=return_logits)
    out5, lg = res if return_logits else (res, None)
    Ho, Wo = q.shape[-2:]
    out = out5.permute(0, 2, 3, 1, 4).reshape(B, Ho, Wo, C).permute(0, 3, 1, 2).float().cpu()
    torch.cuda.synchronize()
    return (out, lg.cpu()) if return_logits else out


def _load_model(dev, params, **kw):
    from naf_amd import NAF
    m = NAF(**kw).eval()
    m.load_state_dict(params, strict=True)
    return m.to(dev)


def assert_close(got, ref, atol, rtol, what=""):
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max err "
                                 f"{float(err.max()):.4e} at {np.unravel_index(int(err.argmax()), err.shape)}; "
                                 f"ref absmax {float(ref.abs().max()):.3f}")


# ---- rope tables / rope+pool / pack -----------------------------------------------------------------
def test_rope_tables(dev):
    from naf_amd import ops
    per = O.rope_periods(256, 4, 100.0)
    ty, tx = ops.rope_tables(per.to(dev), 37, 50)
    ang = O.rope_angles(37, 50, per).view(37, 50, 64)
    assert (ty[:, 0].cpu() - torch.cos(ang[:, 0, :16])).abs().max() < 2e-6
    assert (ty[:, 1].cpu() - torch.sin(ang[:, 0, :16])).abs().max() < 2e-6
    assert (tx[:, 0].cpu() - torch.cos(ang[0, :, 16:32])).abs().max() < 2e-6
    assert (tx[:, 1].cpu() - torch.sin(ang[0, :, 16:32])).abs().max() < 2e-6


@pytest.mark.parametrize("shape,heads,lr,fmt", [
    ((1, 256, 32, 48), 4, (8, 12), "nhwc_f32"),       # integer ratio, vector path
    ((2, 256, 23, 30), 4, (5, 7), "nhwc_bf16"),       # overlapping pool windows (Ho % h != 0)
    ((2, 256, 64, 96), 4, (4, 6), "nhwc_bf16"),       # 16 x 16 cells, bf16 channels-last (the forward's case)
    ((1, 256, 56, 70), 4, (4, 5), "nhwc_bf16"),       # 14 x 14 cells: the last trip of the pixel walk is partial
    ((1, 64, 12, 10), 1, (12, 10), "nchw_f32"),       # ratio 1, one head of 64, strided (NCHW) input
    ((1, 24, 9, 8), 2, (3, 4), "nchw_f32"),           # Dh = 12: scalar path
    ((1, 96, 16, 16), 1, (4, 4), "nhwc_f32"),         # Dh = 96 (denoising dims)
])
def test_rope_pool(dev, shape, heads, lr, fmt):
    from naf_amd import ops
    B, C, H, W = shape
    x = O.hash_normal(shape, 77)
    if "bf16" in fmt:
        x = bf16r(x)
    per = O.rope_periods(C, heads, 100.0)
    ref_q = O.rope(x, per, heads)
    ref_k = O.key_pool(ref_q, lr)
    xd = x.to(dev)
    if "bf16" in fmt:
        xd = xd.to(torch.bfloat16)
    if "nhwc" in fmt:
        xd = xd.contiguous(memory_format=torch.channels_last)
    ty, tx = ops.rope_tables(per.to(dev), H, W)
    for layout in ("head_major", "channels_last"):
        q5, k5 = ops.rope_pool(xd, ty, tx, heads, lr, q_layout=layout)
        q = q5.permute(0, 1, 4, 2, 3).reshape(B, C, H, W).float().cpu()
        k = k5.permute(0, 1, 4, 2, 3).reshape(B, C, *lr).float().cpu()
        assert_close(q, ref_q, 1e-5, 2 ** -8, f"q {fmt} {layout}")       # one bf16 rounding
        assert_close(k, ref_k, 1e-5, 2 ** -8, f"k {fmt} {layout}")
    # keys only (the forward's call: queries are rotated on load by the attention kernel); bf16 channels-last input takes
    # the lean rope_pool_keys_kernel, which must give the SAME bits as the general kernel's keys
    none_q, k5b = ops.rope_pool(xd, ty, tx, heads, lr, write_q=False)
    assert none_q is None and torch.equal(k5b, k5), f"keys-only pass differs: {float((k5b.float() - k5.float()).abs().max())}"


def test_pack_values(dev):
    from naf_amd import ops
    v = O.hash_normal((2, 50, 7, 9), 5)
    for src in (v.to(dev), v.to(dev).to(torch.bfloat16), v.to(dev).permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)):
        vp = ops.pack_values(src)
        assert vp.shape == (2, 7, 9, 50) and vp.dtype == torch.bfloat16
        assert torch.equal(vp.cpu(), src.permute(0, 2, 3, 1).to(torch.bfloat16).cpu())


# ---- fused conv stem ---------------------------------------------------------------------------------
@pytest.mark.parametrize("ks", [1, 3])
@pytest.mark.parametrize("shape", [(1, 20, 24), (2, 37, 70), (1, 64, 32)])
def test_stem_conv0(dev, ks, shape):
    """Conv2d(3 -> 128, reflect) + GroupNorm sums against torch fp32 (output rounded to bf16 once)."""
    import torch.nn.functional as F
    from naf_amd import ops
    B, H, W = shape
    img = O.hash_normal((B, 3, H, W), 61)
    w = O.hash_normal((128, 3, ks, ks), 62, 0.3)
    bias = O.hash_normal((128,), 63, 0.1)
    ref = F.conv2d(F.pad(img, (ks // 2,) * 4, mode="reflect") if ks == 3 else img, w, bias)
    y = torch.empty((B, H, W, 128), dtype=torch.bfloat16, device=dev)
    st = ops.new_stats(B, dev)
    ops.stem_conv0(img.to(dev), w.to(dev), bias.to(dev), y, st)
    got = y.float().cpu().permute(0, 3, 1, 2)
    # one rounding to bf16 (2^-8 |ref| with the tie at a binade's lower end) on top of the kernel's own error before it: products exact
    # in fp32 for the 1x1 layer, carried to 16 bits for the 3x3 layer (<= 2^-15 sum|x||w|, ~3e-5 here: include/naf_hip.h)
    assert_close(got, ref, 1e-5 if ks == 1 else 6e-5, 2 ** -8, f"conv0 k={ks}")
    g = ref.double().view(B, 8, 16, H, W)
    atol = 1e-3 if ks == 1 else 1e-2          # 3x3: the sums of ~1e4 products carried to 16 bits
    assert torch.allclose(ops.stats_total(st)[..., 0].cpu(), g.sum(dim=(2, 3, 4)), rtol=1e-5, atol=atol)
    assert torch.allclose(ops.stats_total(st)[..., 1].cpu(), (g * g).sum(dim=(2, 3, 4)), rtol=1e-5, atol=atol)


@pytest.mark.parametrize("ks", [1, 3])
@pytest.mark.parametrize("shape", [(1, 20, 24), (2, 37, 70), (1, 130, 96)])
def test_stem_conv_layer(dev, ks, shape):
    """One GroupNorm -> SiLU -> Conv layer against torch fp32 on the same bf16 input."""
    import torch.nn.functional as F
    from naf_amd import ops
    B, H, W = shape
    x = bf16r(O.hash_normal((B, 128, H, W), 71) * 1.5 + 0.3)
    w = bf16r(O.hash_normal((128, 128, ks, ks), 72, 1.0 / (11.3 * ks)))
    bias = O.hash_normal((128,), 73, 0.1)
    gw, gb = 1.0 + O.hash_normal((128,), 74, 0.1), O.hash_normal((128,), 75, 0.1)
    a = F.silu(F.group_norm(x, 8, gw, gb, 1e-5))
    a = bf16r(a)                                               # the kernel feeds the MFMA with bf16 activations
    ref = F.conv2d(F.pad(a, (ks // 2,) * 4, mode="reflect") if ks == 3 else a, w, bias)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
    g = x.double().view(B, 8, 16, H, W)
    st_in = ops.stats_from_total(torch.stack([g.sum(dim=(2, 3, 4)), (g * g).sum(dim=(2, 3, 4))], dim=-1).to(dev))
    st_out = ops.new_stats(B, dev)
    y = torch.empty((B, H, W, 128), dtype=torch.bfloat16, device=dev)
    wp = ops.pack_conv_weight(w).to(dev)
    ops.stem_conv(xd, st_in, gw.to(dev), gb.to(dev), 1e-5, wp, bias.to(dev), y, st_out)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert_close(got, ref, 2e-2, 1e-2, f"stem conv k={ks} {shape}")     # bf16 rounding of SiLU(GN(x)) at |x| ~ boundary
    assert float((got - ref).abs().mean()) <= 2e-3
    gr = ref.double().view(B, 8, 16, H, W)
    assert torch.allclose(ops.stats_total(st_out)[..., 0].cpu(), gr.sum(dim=(2, 3, 4)), rtol=2e-3, atol=0.5)
    assert torch.allclose(ops.stats_total(st_out)[..., 1].cpu(), (gr * gr).sum(dim=(2, 3, 4)), rtol=2e-3, atol=0.5)


def test_stem_layers_fuzz_small_and_odd_sizes(dev):
    """Seeded random image sizes from 2x2 up (narrower than a 32-pixel strip, shorter than a step, odd, batched): conv0
    and the GroupNorm -> SiLU -> conv layers of both kernel sizes against torch fp32, as in the two tests above."""
    import torch.nn.functional as F
    from naf_amd import ops
    rng = np.random.RandomState(77)
    sizes = [(2, 2), (2, 33), (3, 3), (5, 64), (33, 2), (64, 31)] + [(int(rng.randint(2, 80)), int(rng.randint(2, 100))) for _ in range(10)]
    for n, (H, W) in enumerate(sizes):
        B = 1 + n % 2
        for ks in (1, 3):
            img = O.hash_normal((B, 3, H, W), 600 + n)
            w0 = O.hash_normal((128, 3, ks, ks), 62, 0.3)
            b0 = O.hash_normal((128,), 63, 0.1)
            ref0 = F.conv2d(F.pad(img, (1,) * 4, mode="reflect") if ks == 3 else img, w0, b0)
            y0 = torch.empty((B, H, W, 128), dtype=torch.bfloat16, device=dev)
            st0 = ops.new_stats(B, dev)
            ops.stem_conv0(img.to(dev), w0.to(dev), b0.to(dev), y0, st0)
            assert_close(y0.float().cpu().permute(0, 3, 1, 2), ref0, 1e-5 if ks == 1 else 6e-5, 2 ** -8, f"conv0 k={ks} {B}x{H}x{W}")
            g0 = ref0.double().view(B, 8, 16, H, W)
            assert torch.allclose(ops.stats_total(st0)[..., 0].cpu(), g0.sum(dim=(2, 3, 4)), rtol=1e-5, atol=1e-3 if ks == 1 else 1e-2)
            # the layer on top of it
            x = bf16r(ref0)
            w = bf16r(O.hash_normal((128, 128, ks, ks), 72, 1.0 / (11.3 * ks)))
            bias = O.hash_normal((128,), 73, 0.1)
            gw, gb = 1.0 + O.hash_normal((128,), 74, 0.1), O.hash_normal((128,), 75, 0.1)
            a = bf16r(F.silu(F.group_norm(x, 8, gw, gb, 1e-5)))
            ref = F.conv2d(F.pad(a, (1,) * 4, mode="reflect") if ks == 3 else a, w, bias)
            gx = x.double().view(B, 8, 16, H, W)
            st_in = ops.stats_from_total(torch.stack([gx.sum(dim=(2, 3, 4)), (gx * gx).sum(dim=(2, 3, 4))], dim=-1).to(dev))
            st_out = ops.new_stats(B, dev)
            y = torch.empty((B, H, W, 128), dtype=torch.bfloat16, device=dev)
            wp = ops.pack_conv_weight(w).to(dev)
            ops.stem_conv(y0, st_in, gw.to(dev), gb.to(dev), 1e-5, wp, bias.to(dev), y, st_out)
            got = y.float().cpu().permute(0, 3, 1, 2)
            assert_close(got, ref, 3e-2, 1.5e-2, f"stem conv k={ks} {B}x{H}x{W}")
            gr = ref.double().view(B, 8, 16, H, W)
            assert torch.allclose(ops.stats_total(st_out)[..., 0].cpu(), gr.sum(dim=(2, 3, 4)), rtol=2e-3, atol=0.5)


@pytest.mark.parametrize("shape", [(1, 20, 24), (2, 45, 67), (1, 160, 64)])
def test_stem_whole_matches_oracle(dev, shape):
    """Both branches, all 5 layers each, against the fp32 oracle conv stem (convolutions.py:67-92)."""
    B, H, W = shape
    p = O.make_params(seed=9)
    m = _load_model(dev, p)
    img = O.hash_normal((B, 3, H, W), 91)
    ref = O.conv_stem(img, p)
    assert m.image_encoder._hip_stem_ok()
    got = m.image_encoder._stem_hip(img.to(dev)).float().cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    assert float(err.mean()) <= 8e-3 and float(err.max()) <= 1.5e-1, (float(err.mean()), float(err.max()))
    m.image_encoder.stem_impl = "torch"
    with torch.no_grad():
        alt = m.image_encoder.guidance(img.to(dev), (H, W)).float().cpu()     # MIOpen bf16 path: same ballpark
    assert float((alt - ref).abs().mean()) <= 2e-2


# ---- attention: staged checks that localise a layout error -------------------------------------------
def test_xna_mfma_uniform_attention_is_window_mean(dev):
    """q = 0 -> uniform weights: out = mean of the clamped window of V (checks gather, masks, V^T reads,
    store layout without any QK dependence)."""
    h, w, d, ksz, C, heads = 9, 11, 4, 7, 128, 4
    q = torch.zeros(1, 256, h * d, w * d)
    k = O.hash_normal((1, 256, h, w), 1)
    v = bf16r(O.hash_normal((1, C, h, w), 2))
    ref = O.xna_lowres(q, k, v, ksz, heads)
    out = run_xna(dev, q, k, v, ksz, heads, path="mfma")
    assert_close(out, ref, 6e-3, 6e-3, "uniform attention")


def test_xna_mfma_constant_values(dev):
    h, w, d, ksz, heads = 8, 8, 4, 5, 4
    q = O.hash_normal((1, 256, h * d, w * d), 3)
    k = O.hash_normal((1, 256, h, w), 4)
    v = torch.arange(128, dtype=torch.float32).view(1, 128, 1, 1).expand(1, 128, h, w).contiguous() / 32.0
    out = run_xna(dev, q, k, v, ksz, heads, path="mfma")
    assert_close(out, v[:, :, :1, :1].expand_as(out), 2e-3, 4e-3, "constant values")


MFMA_CASES = [
    # (B, h, w, dy, dx, ksz, C, heads)
    (1, 8, 12, 4, 4, 7, 128, 4),          # F3-like geometry, every border cell
    (2, 7, 7, 4, 4, 7, 64, 4),            # h == w == k: window is the whole grid (Dv = 16)
    (1, 9, 10, 16, 16, 7, 768, 4),        # G1's cell shape and channel count (Dv = 192)
    (1, 8, 8, 16, 16, 7, 1024, 4),        # G3 channel count (Dv = 256)
    (1, 10, 9, 16, 16, 7, 384, 4),        # P1 channel count (Dv = 96)
    (1, 6, 5, 3, 5, 3, 128, 4),           # 15-query cells: ragged last tile, k = 3
    (1, 6, 7, 2, 4, 5, 256, 4),           # 8-query cells
    (1, 11, 12, 8, 8, 9, 256, 4),         # default kernel 9
    (1, 12, 13, 4, 4, 11, 128, 4),        # kernel 11
    (1, 16, 17, 4, 4, 15, 1024, 4),       # kernel 15 with Dv = 256 (Dv tiles of 128)
    (1, 14, 13, 2, 8, 13, 192, 4),        # kernel 13
    (3, 5, 6, 5, 7, 3, 512, 4),           # odd cell shape 5x7, batch 3
    (1, 6, 7, 14, 14, 5, 192, 4),         # patch-14 cells: row tiles with a partial (14 of 16) tile per row, staged stores
    (1, 5, 5, 3, 30, 3, 256, 4),          # 30-pixel rows: a full and a 14-pixel tile per row
    (1, 9, 9, 7, 15, 9, 1024, 4),         # 15-pixel rows, Dv = 256 (unstaged stores)
    (1, 7, 8, 14, 28, 7, 128, 4),         # 14 x 28 cells
]


@pytest.mark.parametrize("case", MFMA_CASES)
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_xna_mfma_matches_oracle(dev, case, out_dtype):
    B, h, w, dy, dx, ksz, C, heads = case
    seed = sum(case)
    q = bf16r(O.hash_normal((B, 256, h * dy, w * dx), seed + 1))
    k = bf16r(O.hash_normal((B, 256, h, w), seed + 2))
    v = bf16r(O.hash_normal((B, C, h, w), seed + 3))
    ref = O.xna_lowres(q, k, v, ksz, heads)
    out = run_xna(dev, q, k, v, ksz, heads, out_dtype=out_dtype, path="mfma")
    tol = 6e-3 if out_dtype == torch.float32 else 1.2e-2
    assert_close(out, ref, tol, tol, f"mfma {case} {out_dtype}")


def test_xna_mfma_peaked_softmax(dev):
    """Large logits (|s| ~ 40): exercises the max-subtraction path; one key dominates."""
    h, w, d, ksz, heads = 8, 8, 4, 7, 4
    q = bf16r(O.hash_normal((1, 256, h * d, w * d), 9) * 4.0)
    k = bf16r(O.hash_normal((1, 256, h, w), 10) * 4.0)
    v = bf16r(O.hash_normal((1, 128, h, w), 11))
    ref = O.xna_lowres(q, k, v, ksz, heads)
    out = run_xna(dev, q, k, v, ksz, heads, path="mfma")
    assert_close(out, ref, 1e-2, 1e-2, "peaked softmax")
    assert torch.isfinite(out).all()


GENERIC_CASES = [
    # (B, Cq, heads, (Ho, Wo), (h, w), ksz, C)
    (1, 128, 2, (23, 30), (5, 7), 3, 16),      # F4: non-multiple sizes
    (1, 128, 2, (23, 30), (5, 7), 5, 16),
    (2, 64, 1, (24, 20), (24, 20), 5, 3),      # denoising-like: ratio 1, C = 3, one head
    (1, 96, 1, (16, 16), (16, 16), 15, 3),     # kernel 15 at ratio 1, Dq = 96
    (1, 256, 4, (64, 64), (28, 28), 9, 40),    # notebook case 28 -> 64
    (1, 256, 4, (8, 8), (4, 4), 3, 24),        # tiny cells (dy*dx = 4): AUTO picks the table kernel
]


@pytest.mark.parametrize("case", GENERIC_CASES)
def test_xna_generic_matches_oracle(dev, case):
    B, Cq, heads, (Ho, Wo), (h, w), ksz, C = case
    q = bf16r(O.hash_normal((B, Cq, Ho, Wo), 41))
    k = bf16r(O.hash_normal((B, Cq, h, w), 42))
    v = bf16r(O.hash_normal((B, C, h, w), 43))
    ref, ref_lg = O.xna(q, k, v, ksz, heads, return_logits=True)
    out, lg = run_xna(dev, q, k, v, ksz, heads, path="auto", return_logits=True)
    assert_close(out, ref, 2e-5, 2e-5, f"generic out {case}")
    assert_close(lg, ref_lg, 2e-5, 2e-5, f"generic logits {case}")


UNION_CASES = [
    # (B, heads, (Ho, Wo), (h, w), ksz, C): the table-driven MFMA kernel (non-integer ratios, ratio 1, small cells)
    (1, 4, (64, 64), (28, 28), 9, 384),        # notebook case 28 -> 64 (ratio 2.29, repeated taps)
    (1, 4, (128, 128), (28, 28), 9, 384),      # 28 -> 128 (ratio 4.57)
    (1, 4, (100, 150), (37, 37), 9, 768),      # ratio 2.7 x 4.05, Wo not a multiple of 16
    (2, 4, (23, 30), (5, 7), 3, 64),           # F4 sizes
    (1, 4, (23, 30), (5, 7), 5, 64),
    (1, 4, (40, 48), (40, 48), 7, 384),        # ratio 1: plain neighbourhood attention
    (1, 4, (33, 47), (33, 47), 15, 64),        # ratio 1, largest window, ragged tiles
    (1, 4, (32, 32), (16, 16), 7, 768),        # 2x2 cells
    (1, 4, (36, 60), (12, 20), 5, 1024),       # 3x3 cells, Dv = 256
    (1, 4, (64, 64), (16, 16), 7, 128),        # 4x4 cells
    (1, 4, (96, 80), (12, 10), 7, 192),        # 8x8 cells
    (1, 4, (70, 84), (5, 6), 5, 256),          # 14x14 cells (patch 14)
    (1, 4, (300, 40), (20, 17), 11, 64),       # ratio 15 x 2.35, kernel 11
    (1, 2, (50, 50), (21, 21), 13, 2048),      # Dv = 1024: four channel chunks
]


@pytest.mark.parametrize("case", UNION_CASES)
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_xna_union_matches_oracle(dev, case, out_dtype):
    """The MFMA kernel that follows the index tables (repeated taps carry multiplicities) against the oracle's
    dilated hi-res formulation.  Tolerance as for the cell kernel (P is rounded to bf16 before the PV product)."""
    from naf_amd import ops
    B, heads, (Ho, Wo), (h, w), ksz, C = case
    seed = 7 * Ho + Wo + ksz
    q = bf16r(O.hash_normal((B, 64 * heads, Ho, Wo), seed + 1))
    k = bf16r(O.hash_normal((B, 64 * heads, h, w), seed + 2))
    v = bf16r(O.hash_normal((B, C, h, w), seed + 3))
    ref = O.xna(q, k, v, ksz, heads)
    q5, k5 = to5(q, heads).to(dev), to5(k, heads).to(dev)
    v5 = ops.pack_values(v.to(dev)).view(B, h, w, heads, C // heads).permute(0, 3, 1, 2, 4)
    assert ops.xna_select(q5, k5, v5, ksz, path="union") == "union"
    out = run_xna(dev, q, k, v, ksz, heads, out_dtype=out_dtype, path="union")
    tol = 6e-3 if out_dtype == torch.float32 else 1.2e-2
    assert_close(out, ref, tol, tol, f"union {case} {out_dtype}")


def test_union_and_generic_agree_on_a_large_problem(dev):
    """Two independent table-driven kernels, non-integer ratio, more workgroups than CUs."""
    heads, C, ksz = 4, 384, 9
    q = bf16r(O.hash_normal((1, 256, 200, 333), 61))
    k = bf16r(O.hash_normal((1, 256, 37, 41), 62))
    v = bf16r(O.hash_normal((1, C, 37, 41), 63))
    a = run_xna(dev, q, k, v, ksz, heads, path="union")
    b = run_xna(dev, q, k, v, ksz, heads, path="generic")
    assert_close(a, b, 6e-3, 6e-3, "union vs generic")


ROWS_CASES = [
    # (B, heads, Dq, (Ho, Wo), (h, w), ksz, C): the row-streaming MFMA kernel (other head dims, few value channels)
    (2, 1, 64, (24, 20), (24, 20), 5, 3),        # F6 geometry: denoising-like, ratio 1, C = 3
    (1, 1, 96, (33, 47), (33, 47), 15, 3),       # denoising.py:213 defaults: one head of 96, window 15
    (1, 1, 512, (20, 40), (20, 40), 15, 3),      # dim 512
    (1, 2, 128, (64, 64), (28, 28), 9, 10),      # non-integer ratio, Dv = 5
    (1, 4, 64, (23, 30), (5, 7), 3, 24),         # F4 sizes, Dv = 6
    (1, 1, 192, (50, 70), (10, 14), 7, 64),      # 5x5 cells, Dv = 64 (four channel tiles)
    (1, 2, 256, (40, 33), (20, 11), 7, 34),      # Dv = 17: a full and a one-channel tile
    (1, 1, 384, (18, 18), (18, 18), 13, 48),
]


@pytest.mark.parametrize("case", ROWS_CASES)
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_xna_rows_matches_oracle(dev, case, out_dtype):
    from naf_amd import ops
    B, heads, Dq, (Ho, Wo), (h, w), ksz, C = case
    seed = 11 * Ho + Wo + ksz + Dq
    q = bf16r(O.hash_normal((B, Dq * heads, Ho, Wo), seed + 1) * (64.0 / Dq) ** 0.25)
    k = bf16r(O.hash_normal((B, Dq * heads, h, w), seed + 2) * (64.0 / Dq) ** 0.25)
    v = bf16r(O.hash_normal((B, C, h, w), seed + 3))
    ref = O.xna(q, k, v, ksz, heads)
    q5, k5 = to5(q, heads).to(dev), to5(k, heads).to(dev)
    v5 = ops.pack_values(v.to(dev)).view(B, h, w, heads, C // heads).permute(0, 3, 1, 2, 4)
    assert ops.xna_select(q5, k5, v5, ksz) == "rows"                  # what AUTO picks for these shapes
    out = run_xna(dev, q, k, v, ksz, heads, out_dtype=out_dtype, path="rows")
    tol = 6e-3 if out_dtype == torch.float32 else 1.2e-2
    assert_close(out, ref, tol, tol, f"rows {case} {out_dtype}")


def test_rows_fuzz_against_generic(dev):
    from naf_amd import ops
    rng = np.random.RandomState(99)
    done = 0
    for _ in range(300):
        ksz = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 15]))
        h, w = int(rng.randint(max(ksz, 2), 36)), int(rng.randint(max(ksz, 2), 36))
        ry, rx = rng.uniform(1.0, 12.0) ** rng.uniform(0.0, 1.0), rng.uniform(1.0, 12.0) ** rng.uniform(0.0, 1.0)
        Ho, Wo = max(h, int(h * ry)), max(w, int(w * rx))
        if Ho * Wo > 120 * 120 or ksz * (Ho // h) > Ho or ksz * (Wo // w) > Wo:
            continue
        heads, Dq = int(rng.choice([1, 2])), int(rng.choice([64, 96, 128, 192, 256, 384, 512]))
        Dv = int(rng.randint(1, 65))
        q = (torch.randn(1, heads, Ho, Wo, Dq, device=dev) * (64.0 / Dq) ** 0.25).to(torch.bfloat16)
        k = (torch.randn(1, heads, h, w, Dq, device=dev) * (64.0 / Dq) ** 0.25).to(torch.bfloat16)
        v = torch.randn(1, h, w, heads, Dv, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        a = ops.xna_forward(q, k, v, ksz, out_dtype=torch.float32, path="rows")
        b = ops.xna_forward(q, k, v, ksz, out_dtype=torch.float32, path="generic")
        err = (a - b).abs().max().item()
        assert torch.isfinite(a).all() and err <= 6e-3 + 6e-3 * b.abs().max().item(), (h, w, Ho, Wo, ksz, Dq, Dv, heads, err)
        done += 1
        if done >= 50:
            break
    assert done >= 35


def test_union_fuzz_against_generic(dev):
    """Seeded random geometries (ratios 1 .. 20 per axis, every window size, ragged widths, several channel counts):
    the table-driven MFMA kernel against the independent scalar table kernel on the same bf16 inputs."""
    from naf_amd import ops
    rng = np.random.RandomState(1234)
    done = 0
    for _ in range(400):
        ksz = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
        h, w = int(rng.randint(ksz, 40)), int(rng.randint(ksz, 40))
        ry, rx = rng.uniform(1.0, 20.0) ** rng.uniform(0.3, 1.0), rng.uniform(1.0, 20.0) ** rng.uniform(0.3, 1.0)
        Ho, Wo = max(h, int(h * ry)), max(w, int(w * rx))
        if Ho * Wo > 160 * 160 or ksz * (Ho // h) > Ho or ksz * (Wo // w) > Wo:
            continue
        heads = int(rng.choice([1, 2, 4]))
        C = heads * int(rng.choice([16, 32, 48, 96, 192, 256, 512]))
        q = torch.randn(1, heads, Ho, Wo, 64, device=dev).to(torch.bfloat16)
        k = torch.randn(1, heads, h, w, 64, device=dev).to(torch.bfloat16)
        v = torch.randn(1, h, w, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        a = ops.xna_forward(q, k, v, ksz, out_dtype=torch.float32, path="union")
        b = ops.xna_forward(q, k, v, ksz, out_dtype=torch.float32, path="generic")
        err = (a - b).abs().max().item()
        assert torch.isfinite(a).all() and err <= 6e-3 + 6e-3 * b.abs().max().item(), (h, w, Ho, Wo, ksz, C, heads, err)
        done += 1
        if done >= 60:
            break
    assert done >= 40


def test_cell_kernels_fuzz_against_generic(dev):
    """Seeded random integer-ratio geometries (cells 2x4 .. 20x32, all windows, Dv 16 .. 256, bf16 / fp32 output): the cell
    and sliding-window kernels against the scalar table kernel."""
    from naf_amd import ops
    rng = np.random.RandomState(4321)
    done = 0
    for _ in range(400):
        ksz = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
        h, w = int(rng.randint(ksz, 24)), int(rng.randint(ksz, 24))
        dy, dx = int(rng.randint(1, 21)), int(rng.choice([2, 4, 5, 8, 14, 16, 16, 16, 32]))
        Ho, Wo = h * dy, w * dx
        if dy * dx < 8 or Ho * Wo > 200 * 200:
            continue
        heads = int(rng.choice([1, 2, 4]))
        C = heads * int(rng.choice([16, 32, 64, 96, 128, 192, 256]))
        od = torch.float32 if rng.rand() < 0.5 else torch.bfloat16
        q = torch.randn(1, heads, Ho, Wo, 64, device=dev).to(torch.bfloat16)
        k = torch.randn(1, heads, h, w, 64, device=dev).to(torch.bfloat16)
        v = torch.randn(1, h, w, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        a = ops.xna_forward(q, k, v, ksz, out_dtype=od, path="mfma").float()
        b = ops.xna_forward(q, k, v, ksz, out_dtype=torch.float32, path="generic")
        tol = 6e-3 if od == torch.float32 else 1.2e-2
        err = (a - b).abs().max().item()
        assert torch.isfinite(a).all() and err <= tol + tol * b.abs().max().item(), (h, w, dy, dx, ksz, C, heads, od, err)
        done += 1
        if done >= 60:
            break
    assert done >= 40


def test_generic_and_mfma_agree(dev):
    """Two independent kernels on an integer-ratio problem."""
    h, w, d, ksz, heads, C = 10, 9, 8, 7, 4, 192
    q = bf16r(O.hash_normal((1, 256, h * d, w * d), 51))
    k = bf16r(O.hash_normal((1, 256, h, w), 52))
    v = bf16r(O.hash_normal((1, C, h, w), 53))
    a = run_xna(dev, q, k, v, ksz, heads, path="mfma")
    b = run_xna(dev, q, k, v, ksz, heads, path="generic")
    assert_close(a, b, 6e-3, 6e-3, "mfma vs generic")


def test_xna_errors_mirror_reference(dev):
    from naf_amd import ops
    q = torch.zeros(1, 4, 20, 20, 64, dtype=torch.bfloat16, device=dev)
    k = torch.zeros(1, 4, 5, 5, 64, dtype=torch.bfloat16, device=dev)
    v = torch.zeros(1, 4, 5, 5, 16, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError, match="exceeds"):       # NATTEN: kernel*dilation > extent
        ops.xna_forward(q, k, v, 7)
    with pytest.raises(ValueError, match="odd"):
        ops.xna_forward(q, k, v, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.xna_forward(q.cpu(), k.cpu(), v.cpu(), 3)


# ---- golden vectors from the imported reference: attention-only and whole forward -----------------------
def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_golden_F3_attention(dev, golden_dir):
    g = _g(golden_dir, "F3_xna_d4_k7")
    q = O.hash_normal(tuple(g["q_shape"]), int(g["q_seed"]))
    k = O.hash_normal((1, 256, *g["lr"]), int(g["k_seed"]))
    v = O.hash_normal((1, int(g["C"]), *g["lr"]), int(g["v_seed"]))
    out, lg = run_xna(dev, q, k, v, 7, 4, path="auto", return_logits=True)       # table kernel (logits)
    # inputs are rounded to bf16 inside: logits of O(3) with 64-term dots -> abs 4e-2
    assert_close(lg, torch.from_numpy(g["logits"]), 5e-2, 1e-2, "F3 logits vs reference")
    assert_close(out, torch.from_numpy(g["out"]), 4e-2, 2e-2, "F3 out vs reference")
    # Dv = 6 is not MFMA-eligible; an MFMA-eligible variant of the same geometry is covered above


def _forward_stats(got, ref):
    err = (got - ref).abs()
    return float(err.max()), float(err.mean())


def test_golden_F5_full_forward_P1(dev, golden_dir):
    """BASELINE configs[0] end to end against the reference's output (fp32 features -> fp32 output)."""
    g = _g(golden_dir, "F5_full_P1")
    p = O.make_params(seed=int(g["param_seed"]))
    m = _load_model(dev, p, kernel_size=int(g["k"]))
    img = O.hash_normal((1, 3, 224, 224), int(g["image_seed"]))
    ft = O.hash_normal((1, 384, 14, 14), int(g["feat_seed"]))
    out = m(img.to(dev), ft.to(dev), (224, 224))
    assert out.shape == (1, 384, 224, 224) and out.dtype == torch.float32
    out = out.float().cpu()
    oy, ox = g["offset"]
    st = int(g["stride"])
    ref = torch.from_numpy(g["sample"])
    got = out[:, :, oy::st, ox::st]
    assert_close(got, ref, 2e-2, 1e-2, "F5 strided sample")
    assert _forward_stats(got, ref)[1] <= 6e-3
    assert_close(out[:, ::48, :2, :], torch.from_numpy(g["top_rows"]), 2e-2, 1e-2, "F5 top rows")
    assert_close(out[:, ::48, :, -2:], torch.from_numpy(g["left_cols"]), 2e-2, 1e-2, "F5 right cols")
    assert (out.mean(dim=(0, 2, 3)) - torch.from_numpy(g["ch_mean"])).abs().max() <= 5e-3
    # bf16 features -> bf16 output, same values within one more rounding
    out_b = m(img.to(dev), ft.to(dev).to(torch.bfloat16), [224, 224])
    assert out_b.dtype == torch.bfloat16
    assert_close(out_b.float().cpu()[:, :, oy::st, ox::st], ref, 2e-2, 1e-2, "F5 bf16 I/O")      # measured 2.7e-3 (r04_tolerance_budget.txt)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_golden_F9_noninteger_ratio(dev, golden_dir, tag):
    """Whole forward at non-integer ratios against the REFERENCE's output (notebook geometry 28^2 -> 64^2; 37^2 ->
    100x150): the table-driven MFMA kernel with the index tables naf_forward builds on the device."""
    from naf_amd import ops
    g = _g(golden_dir, "F9_noninteger_ratio")
    p = O.make_params(seed=int(g["param_seed"]))
    m = _load_model(dev, p, kernel_size=int(g["k"]))
    H, W, h, w, C, iseed = (int(v) for v in g[f"{tag}_shape"])
    img = O.hash_normal((1, 3, H, W), iseed).to(dev)
    ft = O.hash_normal((1, C, h, w), iseed + 1).to(dev)
    assert m._forward_plan(img, ft, (H, W)) is not None
    q5 = torch.empty(1, 4, H, W, 64, dtype=torch.bfloat16, device=dev)
    k5 = torch.empty(1, 4, h, w, 64, dtype=torch.bfloat16, device=dev)
    v5 = torch.empty(1, h, w, 4, C // 4, dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
    assert ops.xna_select(q5, k5, v5, int(g["k"])) == "union"
    out = m(img, ft, (H, W)).float().cpu()
    got = out[:, :, 1::2, ::2] if tag == "a" else out[:, ::4, 1::2, ::3]
    ref = torch.from_numpy(g[f"{tag}_sample"])
    assert_close(got, ref, 2e-2, 1e-2, f"F9{tag} strided sample vs reference")
    assert _forward_stats(got, ref)[1] <= 6e-3


@pytest.mark.parametrize("feat_dtype", [torch.float32, torch.bfloat16])
def test_golden_F10_patch14(dev, golden_dir, feat_dtype):
    """14-pixel cells (patch-14 backbones) end to end against the REFERENCE's output: the cell kernel's row tiles with a
    partial last tile, queries rotated on load."""
    from naf_amd import ops
    g = _g(golden_dir, "F10_patch14")
    p = O.make_params(seed=int(g["param_seed"]))
    m = _load_model(dev, p, kernel_size=int(g["k"]))
    H, W, h, w, C = (int(v) for v in g["shape"])
    img = O.hash_normal((1, 3, H, W), int(g["image_seed"])).to(dev)
    ft = O.hash_normal((1, C, h, w), int(g["feat_seed"])).to(dev).to(feat_dtype)
    plan = m._forward_plan(img, ft, (H, W))
    assert plan is not None
    q_raw = torch.empty(1, 4, H, W, 64, dtype=torch.bfloat16, device=dev)
    assert ops.xna_rope_fusable(q_raw, (h, w), C // 4, int(g["k"]), m.image_encoder.rope.tables(H, W))
    out = m(img, ft, (H, W)).float().cpu()
    got, ref = out[:, ::2, ::3, 1::3], torch.from_numpy(g["sample"])
    assert_close(got, ref, 2e-2, 1e-2, "F10 strided sample vs reference")     # measured 3.3e-3 (fp32 features) / 3.7e-3 (bf16)
    assert _forward_stats(got, ref)[1] <= 6e-3


def test_golden_F6_denoise_like(dev, golden_dir):
    g = _g(golden_dir, "F6_denoise_d1")
    p = O.make_params(dim=int(g["dim"]), heads_rope=1, seed=int(g["param_seed"]))
    m = _load_model(dev, p, dim=int(g["dim"]), heads_attn=1, heads_rope=1, kernel_size=int(g["k"]))
    shp = tuple(g["shape"])
    img, ft = O.hash_normal(shp, int(g["image_seed"])), O.hash_normal(shp, int(g["feat_seed"]))
    out, lg = m(img.to(dev), ft.to(dev), torch.Size(shp[-2:]), return_weights=True)
    assert lg.shape == tuple(g["logits"].shape)
    # One head of 64 dims, a 5x5 window, three value channels: the softmax is peaked (top weight 0.86 at the worst element) and the
    # output follows single keys.  Measured (tools/f6_error_probe.py, profiles/r03_f6_error.txt): max error 3.6e-2, 2 of 2880
    # elements outside SURVEY 8c's 2e-2 + 1e-2 |ref|; the attention kernels alone, fed the oracle's fp32 guidance, stay within
    # 6.3e-3 of the golden output -- the excess is the bf16-activation stem moving logits of magnitude up to 37 by up to 0.13
    # at those elements (0.34 anywhere).  Round 2 passed this at 1e-1 + 3e-2 |ref|.
    ref, ref_lg = torch.from_numpy(g["out"]), torch.from_numpy(g["logits"])
    got = out.float().cpu()
    assert_close(got, ref, 4.5e-2, 1e-2, "F6 out")
    err = (got - ref).abs()
    assert float((err > 2e-2 + 1e-2 * ref.abs()).float().mean()) <= 3e-3, "F6: more than 0.3 % of the outputs outside 2e-2 + 1e-2 |ref|"
    assert _forward_stats(got, ref)[1] <= 3e-3
    assert_close(lg.cpu(), ref_lg, 1e-1, 3e-2, "F6 logits (pre-softmax, scaled, |ref| up to 37)")
    lerr = (lg.cpu() - ref_lg).abs()
    assert float((lerr > 2e-2 + 1e-2 * ref_lg.abs()).float().mean()) <= 5e-3 and float(lerr.mean()) <= 8e-3
    # the attention on the ORACLE's guidance (fp32 stem, bf16 q/k/v contract): within 1e-2 of the reference's output
    from naf_amd import ops
    with torch.no_grad():
        xq = O.image_encoder(img, shp[-2:], p, 1)
        kk = O.key_pool(xq, shp[-2:])
    t5 = lambda t: t.view(t.shape[0], 1, t.shape[1], *t.shape[-2:]).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16).to(dev)
    o2 = ops.xna_forward(t5(xq), t5(kk), t5(ft), int(g["k"]), out_dtype=torch.float32)
    o2 = o2.permute(0, 1, 4, 2, 3).reshape(ref.shape).float().cpu()
    assert_close(o2, ref, 1e-2, 1e-2, "F6: HIP attention on the oracle's guidance")


def test_golden_F7_preshrink_and_pool(dev, golden_dir):
    g = _g(golden_dir, "F7_preshrink_pool")
    p = O.make_params(dim=int(g["dim"]), heads_rope=2, seed=int(g["param_seed"]))
    m = _load_model(dev, p, dim=int(g["dim"]), heads_attn=2, heads_rope=2, kernel_size=int(g["k"]))
    ft = O.hash_normal(tuple(g["feat_shape"]), int(g["feat_seed"]))
    for tag in ("a", "b"):
        img = O.hash_normal(tuple(g[f"image_shape_{tag}"]), int(g[f"image_seed_{tag}"]))
        out = m(img.to(dev), ft.to(dev), tuple(int(s) for s in g[f"out_size_{tag}"]))
        assert_close(out.float().cpu(), torch.from_numpy(g[f"out_{tag}"]), 2e-2, 1e-2, f"F7{tag}")


@pytest.mark.parametrize("B,C,lr,out_sz,ksz,out_dtype", [
    (1, 64, (8, 8), (128, 128), 7, torch.bfloat16),        # d = 16: one row tile per cell row
    (2, 128, (6, 5), (192, 160), 5, torch.float32),        # d = 32: two tiles per cell row, non-square grid
    (1, 64, (12, 12), (96, 192), 3, torch.bfloat16),       # dy = 8, dx = 16
    (1, 192, (6, 7), (84, 98), 5, torch.bfloat16),         # patch 14: partial row tiles (14 of 16 lanes)
    (1, 64, (5, 5), (35, 150), 3, torch.float32),          # dx = 30: a full and a 14-pixel tile per row
])
def test_rotate_on_load_equals_materialised_queries(dev, B, C, lr, out_sz, ksz, out_dtype):
    """naf_xna_fwd(rope_tab_*) on un-rotated guidance == naf_rope_pool_fwd queries + plain naf_xna_fwd, bit for
    bit (same rotation arithmetic, same bf16 rounding), and both match the oracle's rope + attention."""
    from naf_amd import ops
    heads, Dq = 4, 64
    x = bf16r(O.hash_normal((B, heads * Dq, *out_sz), 310))
    v = bf16r(O.hash_normal((B, C, *lr), 311))
    per = O.rope_periods(heads * Dq, heads, 100.0)
    xd = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ty, tx = ops.rope_tables(per.to(dev), *out_sz)
    q_mat, k5 = ops.rope_pool(xd, ty, tx, heads, lr)
    none_q, k5b = ops.rope_pool(xd, ty, tx, heads, lr, write_q=False)
    assert none_q is None and torch.equal(k5, k5b)
    q_raw = xd.permute(0, 2, 3, 1).unflatten(3, (heads, Dq)).permute(0, 3, 1, 2, 4)
    assert ops.xna_rope_fusable(q_raw, lr, C // heads, ksz, (ty, tx), out_dtype=out_dtype)
    vp = ops.pack_values(v.to(dev))
    v5 = vp.view(B, *lr, heads, C // heads).permute(0, 3, 1, 2, 4)
    a = ops.xna_forward(q_mat, k5, v5, ksz, out_dtype=out_dtype, path="mfma")
    b = ops.xna_forward(q_raw, k5, v5, ksz, out_dtype=out_dtype, path="mfma", rope_tables=(ty, tx))
    assert torch.equal(a, b), f"rotate-on-load differs from materialised queries: {float((a.float() - b.float()).abs().max())}"
    ref_q = bf16r(O.rope(x, per, heads))
    ref_k = bf16r(O.key_pool(O.rope(x, per, heads), lr))
    ref = O.xna_lowres(ref_q, ref_k, v, ksz, heads)
    got = b.permute(0, 1, 4, 2, 3).reshape(B, C, *out_sz).float().cpu()
    assert_close(got, ref, 2e-2, 2e-2, "rotate-on-load vs oracle")


@pytest.mark.parametrize("lr,out_sz,ksz", [((8, 8), (128, 128), 7), ((6, 5), (24, 30), 3), ((12, 12), (96, 192), 9)])
def test_return_weights_from_the_mfma_kernel(dev, lr, out_sz, ksz):
    """return_weights (attentions.py:27-28: scaled pre-softmax scores, row-major window order) straight from the MFMA
    cell kernel: row-tile path (d = 16), generic tile path (d = 4 x 6) and a 9x9 window with pad slots."""
    heads, C = 4, 64
    q = bf16r(O.hash_normal((1, 256, *out_sz), 401))
    k = bf16r(O.hash_normal((1, 256, *lr), 402))
    v = bf16r(O.hash_normal((1, C, *lr), 403))
    ref, ref_lg = O.xna(q, k, v, ksz, heads, return_logits=True)
    out, lg = run_xna(dev, q, k, v, ksz, heads, path="mfma", return_logits=True)
    assert_close(lg, ref_lg, 2e-5, 2e-5, "mfma logits")
    assert_close(out, ref, 6e-3, 6e-3, "mfma out (with logits)")


def test_rotate_on_load_refused_when_tiles_straddle_rows(dev):
    from naf_amd import ops
    from naf_amd._lib import NafHipError
    heads, Dq = 4, 64
    xd = torch.zeros((1, 40, 40, heads * Dq), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2)   # d = 8
    per = O.rope_periods(heads * Dq, heads, 100.0)
    ty, tx = ops.rope_tables(per.to(dev), 40, 40)
    q_raw = xd.permute(0, 2, 3, 1).unflatten(3, (heads, Dq)).permute(0, 3, 1, 2, 4)
    assert not ops.xna_rope_fusable(q_raw, (5, 5), 16, 3, (ty, tx))
    _, k5 = ops.rope_pool(xd, ty, tx, heads, (5, 5), write_q=False)
    v5 = torch.zeros((1, 5, 5, heads, 16), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
    with pytest.raises(NafHipError):
        ops.xna_forward(q_raw, k5, v5, 3, rope_tables=(ty, tx))


def test_oracle_on_the_device_equals_the_oracle_on_the_host(dev):
    """The oracle is plain torch code; the largest parity cases (the 2048^2 and G3 stems, fp64 autograd of wide heads with large
    windows) evaluate that SAME code through ATen's device kernels to keep the suite inside its time cap.  This test holds the
    device evaluation to the host evaluation on this box: the whole forward in fp32 (convolutions, GroupNorm, RoPE, pooling,
    gather, softmax) and the fp64 autograd of the attention."""
    p = O.make_params(seed=97)
    img = O.hash_normal((1, 3, 96, 80), 9701)
    ft = O.hash_normal((1, 64, 6, 5), 9702)
    with torch.no_grad():
        host = O.naf_forward(p, img, ft, (96, 80), kernel_size=5)
        stem_h = O.conv_stem(img, p)
        pd = {k: v.to(dev) for k, v in p.items()}
        devc = O.naf_forward(pd, img.to(dev), ft.to(dev), (96, 80), kernel_size=5).cpu()
        stem_d = O.conv_stem(img.to(dev), pd).cpu()
    assert float((stem_d - stem_h).abs().max()) <= 2e-5 * max(1.0, float(stem_h.abs().max())), float((stem_d - stem_h).abs().max())
    assert float((devc - host).abs().max()) <= 2e-5, float((devc - host).abs().max())
    q = bf16r(O.hash_normal((1, 128, 48, 64), 9703))
    k = bf16r(O.hash_normal((1, 128, 6, 8), 9704))
    v = bf16r(O.hash_normal((1, 64, 6, 8), 9705))
    g = bf16r(O.hash_normal((1, 64, 48, 64), 9706))
    h3 = O.xna_backward(q, k, v, g, 5, 2)
    d3 = O.xna_backward(q.to(dev), k.to(dev), v.to(dev), g.to(dev), 5, 2)
    for a, b, name in zip(d3, h3, ("dq", "dk", "dv")):
        assert float((a.cpu() - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max())), name


@pytest.mark.parametrize("B,C,lr,out_sz,ksz", [
    (1, 256, (8, 8), (128, 128), 7),       # Dv = 64, d = 16: four row tiles per wave round, full rounds
    (2, 128, (5, 6), (40, 96), 3),         # Dv = 32, d = (8, 16): 8 tiles per cell, non-square grid
    (1, 768, (7, 7), (7, 112), 7),         # Dv = 192, dy = 1: one tile per cell (three dead waves per round), k = h = w
    (1, 384, (10, 9), (80, 288), 9),       # Dv = 96, 9x9 window (pad slots), dx = 32
    (1, 512, (10, 11), (160, 176), 9),     # Dv = 128, 9x9: eight-wave kernel with ONE window buffer and K fragments from the LDS
    (1, 768, (10, 12), (80, 192), 9),      # Dv = 192, 9x9 (ViT-B features at the reference's default window): 88-slot P / dS rows, three resident V key tiles
    (1, 1024, (10, 11), (80, 176), 9),     # Dv = 256, 9x9 (DINOv3-L features at the reference's default window): ONE P / dS buffer, two barriers per round, no resident V tile
    (2, 768, (9, 12), (72, 192), 7),       # Dv = 192, 7x7, two images: runs of cells across images and heads (G1's instantiation)
    (2, 1024, (9, 10), (72, 160), 7),      # Dv = 256 at 7x7 (BASELINE's G2 / G3 width): eight-wave kernel, one V key tile from the LDS per round, columns staged in two passes
    (1, 384, (12, 14), (96, 224), 11),     # 11x11 at Dv = 96: the eight-wave kernel (eight key tiles, K fragments one tile at a time)
    (1, 512, (13, 12), (52, 192), 11),     # 11x11 at Dv = 128: ... with ONE P / dS buffer
    (1, 1024, (12, 13), (96, 208), 11),    # 11x11 window, Dv = 256 (BASELINE's G2 width): the eight-wave kernel in two channel chunks of 128 (dQ accumulated across the launches)
    (1, 512, (13, 14), (104, 224), 13),    # 13x13, Dv = 128: chunks 64 + 64; two rounds per cell
    (1, 768, (13, 14), (26, 224), 13),     # 13x13, Dv = 192: channel chunks 64 x 3 on the eight-wave kernel (dQ of the later launches adds to the first's)
    (1, 512, (15, 16), (30, 256), 15),     # 15x15 (BASELINE configs[2]'s largest window), Dv = 128: chunks 64 + 64 on the eight-wave kernel (P / dS rows of 240 slots)
    (2, 384, (16, 15), (16, 240), 15),     # 15x15, Dv = 96: chunks 64 + 32, one-row cells (three dead waves per round), two images
    # round 6: PARTIAL row tiles (xna_bwd2_kernel PT) -- cell rows that are not a multiple of 16 pixels, the patch-14 backbones' 14 first
    (1, 384, (12, 10), (168, 140), 9),     # ratio 14 at the reference's default window, Dv = 96: 14 of a tile's 16 lanes hold a query (row-streaming kernel until round 5)
    (1, 256, (8, 9), (120, 135), 7),       # ratio 15, Dv = 64
    (2, 768, (9, 10), (126, 280), 9),      # cells of 14 x 28 pixels: two tiles per row, the second with 12 queries; Dv = 192, two images
    (1, 1024, (8, 9), (32, 126), 7),       # ratio (4, 14) at Dv = 256: one V key tile from the LDS per round, dead query waves AND idle lanes
    (1, 128, (5, 6), (150, 180), 5),       # cells of 30 x 30 pixels (16 + 14), Dv = 32
])
def test_xna_backward_matches_oracle(dev, B, C, lr, out_sz, ksz):
    """naf_xna_bwd vs autograd through the oracle's forward, same bf16-rounded q, k, v and output gradient."""
    from naf_amd import ops
    heads = 4
    q = bf16r(O.hash_normal((B, 256, *out_sz), 501))
    k = bf16r(O.hash_normal((B, 256, *lr), 502))
    v = bf16r(O.hash_normal((B, C, *lr), 503))
    dout = bf16r(O.hash_normal((B, C, *out_sz), 504))
    rq, rk, rv = oracle_xna_backward(dev, q, k, v, dout, ksz, heads)
    q5, k5, v5, g5 = (to5(t, heads).to(dev) for t in (q, k, v, dout))
    assert ops.xna_backward_supported(q5, k5, v5, ksz)
    dq, dk, dv = ops.xna_backward(q5, k5, v5, g5, ksz)
    back = lambda t5: t5.permute(0, 1, 4, 2, 3).reshape(t5.shape[0], -1, *t5.shape[2:4]).float().cpu()
    # dS and P pass through bf16 (relative 2^-8) before the contractions; sums of up to d^2 * k^2 terms
    for got, ref, name in ((back(dq), rq, "dq"), (back(dk), rk, "dk"), (back(dv), rv, "dv")):
        scale = float(ref.abs().max())
        err = (got - ref).abs()
        assert float(err.max()) <= 2e-2 * scale + 1e-3 and float(err.mean()) <= 3e-3 * scale + 1e-4, \
            f"{name}: max err {float(err.max()):.3e} mean {float(err.mean()):.3e} (ref max {scale:.3e})"


@pytest.mark.parametrize("C,lr,out_sz,ksz,nchunk", [(1024, (12, 13), (48, 208), 11, 2), (1024, (15, 16), (30, 256), 15, 4), (768, (13, 14), (26, 224), 13, 3)])
def test_chunked_backward_dq_against_the_one_launch_kernel(dev, C, lr, out_sz, ksz, nchunk):
    """wide heads at 11 x 11 and every head at 13 x 13 / 15 x 15 run the cell backward in channel chunks, and dQ -- a sum over
    the chunks -- travels through the caller's bf16 buffer: one rounding PER CHUNK (include/naf_hip.h, naf_xna_bwd_chunk_plan).  The
    row-streaming kernel (path="rows") computes the same dQ in ONE launch with one rounding: against the fp64 oracle the chunked dQ may be
    worse than that by the extra roundings only (measured 1.1-1.3 x; bound: 1.6 x + 5e-4), and the two agree to bf16 accuracy."""
    from naf_amd import ops
    heads = 4
    q = bf16r(O.hash_normal((1, 256, *out_sz), 561))
    k = bf16r(O.hash_normal((1, 256, *lr), 562))
    v = bf16r(O.hash_normal((1, C, *lr), 563))
    dout = bf16r(O.hash_normal((1, C, *out_sz), 564))
    rq, _, _ = oracle_xna_backward(dev, q, k, v, dout, ksz, heads)
    q5, k5, v5, g5 = (to5(t, heads).to(dev) for t in (q, k, v, dout))
    assert ops.xna_backward_select(q5, k5, v5, ksz) == "mfma" and len(ops.xna_backward_chunks(q5, k5, v5, ksz)) == nchunk
    back = lambda t5: t5.permute(0, 1, 4, 2, 3).reshape(t5.shape[0], -1, *t5.shape[2:4]).float().cpu()
    dq_c = back(ops.xna_backward(q5, k5, v5, g5, ksz)[0])
    dq_r = back(ops.xna_backward(q5, k5, v5, g5, ksz, path="rows")[0])
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    e_c, e_r, e_cr = rel(dq_c, rq), rel(dq_r, rq), rel(dq_c, dq_r)
    print("chunked dQ, k %d, %d chunks: %.3e of |dq| from the oracle (one launch: %.3e), %.3e between the two" % (ksz, nchunk, e_c, e_r, e_cr))
    assert e_r <= 6e-3 and e_c <= 1.6 * e_r + 5e-4 and e_cr <= 8e-3, (e_c, e_r, e_cr)


@pytest.mark.parametrize("B,Cq,C,heads,lr,out_sz,ksz", [
    (1, 160, 24, 4, (5, 7), (23, 30), 3),       # non-integer ratio (F4 shapes), heads of 40 dims: irregular neighbourhoods, duplicates
    (1, 80, 3, 1, (12, 10), (12, 10), 5),       # ratio 1, one head of 80 (no matrix-core instantiation), C = 3
    (2, 64, 16, 2, (4, 4), (16, 16), (3, 1)),   # rectangular window, Dq = 32, d = 4 (MFMA backward does not serve it)
])
def test_xna_backward_table_driven_matches_oracle(dev, B, Cq, C, heads, lr, out_sz, ksz):
    """naf_xna_bwd's table-driven kernel (shapes the MFMA cell kernel does not serve) vs autograd through the oracle."""
    from naf_amd import ops
    q = bf16r(O.hash_normal((B, Cq, *out_sz), 531))
    k = bf16r(O.hash_normal((B, Cq, *lr), 532))
    v = bf16r(O.hash_normal((B, C, *lr), 533))
    dout = bf16r(O.hash_normal((B, C, *out_sz), 534))
    rq, rk, rv = O.xna_backward(q, k, v, dout, ksz, heads)
    q5, k5, v5, g5 = (to5(t, heads).to(dev) for t in (q, k, v, dout))
    assert not ops.xna_backward_supported(q5, k5, v5, ksz)
    dq, dk, dv = ops.xna_backward(q5, k5, v5, g5, ksz)
    back = lambda t5: t5.permute(0, 1, 4, 2, 3).reshape(t5.shape[0], -1, *t5.shape[2:4]).float().cpu()
    for got, ref, name, tol in ((back(dq), rq, "dq", 1e-2), (back(dk), rk, "dk", 1e-4), (back(dv), rv, "dv", 1e-4)):
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err <= tol * scale + 1e-5, f"{name}: max err {err:.3e} (ref max {scale:.3e})"   # dq is rounded to bf16


def test_cell_backward_fuzz_against_table_driven_kernel(dev):
    """Seeded random geometries of the MFMA cell backward -- every window 3 .. 15, every Dv (13 x 13 and 15 x 15 in channel chunks), row-tile counts that are not multiples of the
    four tiles of a round (dead query waves), one-row cells, several images and heads -- against the independent scalar table-driven
    kernel (fp32 throughout) on the same bf16 inputs.  Every window runs the wave-specialised eight-wave kernel (xna_bwd2_kernel.h: query
    waves / key waves): 3 ... 11 whole (11 x 11 beyond Dv = 128 in two chunks), 13 x 13 / 15 x 15 in channel chunks of <= 64; the four-wave
    kernel of xna_bwd_kernel.h only behind the A/B knobs NAF_BWD_V1 / NAF_BWD_BIG8=0 / NAF_BWD_CHUNK11=0."""
    from naf_amd import ops
    seed, want = int(os.environ.get("NAF_FUZZ_BWD_SEED", "9753")), int(os.environ.get("NAF_FUZZ_BWD_CASES", "70"))   # campaigns: profiles/r05_fuzz_backward.txt
    rng = np.random.RandomState(seed)
    done = ragged = small = chunked = partial = 0
    for _ in range(8 * want):
        ksz = int(rng.choice([3, 5, 7, 7, 7, 9, 11, 13, 15, 15]))
        h, w = int(rng.randint(ksz, ksz + 6)), int(rng.randint(ksz, ksz + 6))
        dy, dx = int(rng.choice([1, 2, 3, 5, 6, 8, 16])), int(rng.choice([16, 16, 32, 48, 14, 15, 28, 30]))      # (14 ... 30: partial row tiles, windows <= 9)
        Ho, Wo = h * dy, w * dx
        if Ho * Wo > 160 * 400:
            continue
        B, heads = int(rng.choice([1, 2])), int(rng.choice([1, 2, 4]))
        Dv = int(rng.choice([32, 64, 96, 128, 192, 256]))
        q = torch.randn(B, heads, Ho, Wo, 64, device=dev).to(torch.bfloat16)
        k = torch.randn(B, heads, h, w, 64, device=dev).to(torch.bfloat16)
        v = torch.randn(B, h, w, heads, Dv, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        g = torch.randn(B, Ho, Wo, heads, Dv, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        if ops.xna_backward_select(q, k, v, ksz) != "mfma":
            continue
        a = ops.xna_backward(q, k, v, g, ksz)
        b = ops.xna_backward(q, k, v, g, ksz, path="generic")
        for x, y, name in zip(a, b, ("dq", "dk", "dv")):
            scale = float(y.float().abs().max())
            err = float((x.float() - y.float()).abs().max())
            assert bool(torch.isfinite(x.float()).all()) and err <= 2.5e-2 * scale + 1e-3, (name, B, heads, h, w, Ho, Wo, ksz, Dv, err, scale)
        done += 1
        ragged += int((dy * ((dx + 15) // 16)) % 4 != 0)
        partial += int(dx % 16 != 0)           # round 6: rows whose last tile holds fewer than 16 queries (windows <= 9)
        small += int(ksz <= 7)
        chunked += int((ksz == 11 and Dv > 128) or (ksz == 13 and Dv > 64) or (ksz == 15 and Dv > 64))
        if os.environ.get("NAF_FUZZ_BWD_CASES"):
            print("bwd fuzz %d: k %d lr (%d, %d) out (%d, %d) B %d heads %d Dv %d: dq/dk/dv max err / max |ref| %s" % (
                seed, ksz, h, w, Ho, Wo, B, heads, Dv, " ".join("%.2e" % (float((x.float() - y.float()).abs().max()) / (float(y.float().abs().max()) + 1e-30)) for x, y in zip(a, b))))
        if done >= want:
            break
    assert done >= 40 and ragged >= 10 and small >= 20 and chunked >= 3 and partial >= 8, (done, ragged, small, chunked, partial)


@pytest.mark.parametrize("B,heads,lr,d,Dv,ksz", [
    (2, 4, (40, 40), (16, 16), 192, 7),     # G1's instantiation, 320 cell rows x heads: segments of ~10 cells, five runs per workgroup
    (3, 2, (44, 46), (4, 16), 256, 7),      # BASELINE's G2 / G3 width: one window buffer, a V key tile from the LDS, columns staged in two passes
    (2, 4, (36, 50), (8, 16), 96, 9),       # 9 x 9 (the reference's default window), two window buffers
    (2, 4, (34, 38), (6, 32), 128, 9),      # 9 x 9 at Dv = 128: one window buffer, K fragments from the LDS; three rounds per cell
    (2, 4, (33, 40), (4, 16), 192, 9),      # 9 x 9 at Dv = 192 (C = 768): 88-slot P / dS rows, columns staged in two passes
    (2, 4, (35, 37), (2, 32), 256, 9),      # 9 x 9 at Dv = 256 (C = 1024): one P / dS buffer (two barriers per round) AND one window buffer
    (2, 4, (33, 35), (4, 16), 128, 11),     # 11 x 11 at Dv = 128: the widest the eight-wave kernel takes at that window
    (2, 4, (34, 33), (2, 16), 256, 11),     # 11 x 11 at Dv = 256 (BASELINE's G2 width): two channel chunks of 128 on the eight-wave kernel
    (2, 4, (33, 34), (2, 16), 192, 11),     # 11 x 11 at Dv = 192: chunks 96 + 96
    (2, 4, (33, 35), (2, 16), 128, 13),     # 13 x 13: chunks 64 + 64 on the eight-wave kernel (twelve key tiles, 8 of 12 V tiles resident)
    (2, 4, (34, 33), (1, 16), 96, 15),      # 15 x 15 (BASELINE configs[2]'s largest window): chunks 64 + 32, sixteen key tiles, one-row cells
])
def test_cell_backward_walks_several_runs_per_workgroup(dev, B, heads, lr, d, Dv, ksz):
    """The wave-specialised backward launches one resident workgroup per CU and lets it walk runs of cells (xna_bwd2_kernel.h); every other
    backward case in this file has fewer runs than the chip has CUs, i.e. ONE run per workgroup -- as have G1, G2 and a G3 image.  Here
    B x h x heads exceeds 256, so workgroups leave a run (flush of the whole window, windows of another cell row / head / image staged
    from scratch, walker and key-wave bookkeeping re-decoded) several times.  Checked against the independent table-driven scalar kernel
    (fp32 throughout) on the same bf16 inputs."""
    from naf_amd import ops
    h, w = lr
    Ho, Wo = h * d[0], w * d[1]
    assert B * h * heads > 256
    gen = torch.Generator(device=dev).manual_seed(4242 + Dv + ksz)
    q = torch.randn(B, heads, Ho, Wo, 64, device=dev, generator=gen).to(torch.bfloat16)
    k = torch.randn(B, heads, h, w, 64, device=dev, generator=gen).to(torch.bfloat16)
    v = torch.randn(B, h, w, heads, Dv, device=dev, generator=gen).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    g = torch.randn(B, Ho, Wo, heads, Dv, device=dev, generator=gen).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    assert ops.xna_backward_select(q, k, v, ksz) == "mfma"
    a = ops.xna_backward(q, k, v, g, ksz)
    b = ops.xna_backward(q, k, v, g, ksz, path="generic")
    for x, y, name in zip(a, b, ("dq", "dk", "dv")):
        scale = float(y.float().abs().max())
        err = (x.float() - y.float()).abs()
        assert bool(torch.isfinite(x.float()).all()), name
        assert float(err.max()) <= 2.5e-2 * scale + 1e-3 and float(err.mean()) <= 3e-3 * scale + 1e-4, (name, float(err.max()), float(err.mean()), scale)


@pytest.mark.parametrize("B,Cq,C,heads,size,ksz", [
    (1, 96, 3, 1, (12, 10), 5),        # narrower than one 16-query tile
    (1, 96, 3, 1, (40, 52), 15),       # the denoising call's shape class: one head, RGB values, 15x15 window, ragged last tile
    (2, 256, 3, 1, (33, 47), 15),      # NAF(dim = 256), batch 2
    (1, 512, 3, 1, (24, 40), 15),      # NAF(dim = 512): 16 k-steps, 128 accumulator registers, one workgroup per CU
    (1, 128, 16, 2, (20, 36), 9),      # two heads of 64, 8 value channels per head
    (1, 64, 32, 1, (17, 17), 7),       # 32 value channels: two channel tiles
    (1, 192, 24, 1, (15, 31), 15),     # window as tall as the image: every query row sees every key row
    (1, 384, 3, 1, (16, 64), 3),       # smallest window, whole tiles only
])
def test_xna_backward_rows_matches_oracle(dev, B, Cq, C, heads, size, ksz):
    """The row-streaming matrix-core backward (keys and queries on one grid: denoising.py:213,301) vs autograd through the oracle."""
    _rows_backward_case(dev, B, Cq, C, heads, size, size, ksz)


def _rows_backward_case(dev, B, Cq, C, heads, lr, out_sz, ksz):
    from naf_amd import ops
    q = bf16r(O.hash_normal((B, Cq, *out_sz), 541))
    k = bf16r(O.hash_normal((B, Cq, *lr), 542))
    v = bf16r(O.hash_normal((B, C, *lr), 543))
    dout = bf16r(O.hash_normal((B, C, *out_sz), 544))
    rq, rk, rv = oracle_xna_backward(dev, q, k, v, dout, ksz, heads)
    q5, k5, v5, g5 = (to5(t, heads).to(dev) for t in (q, k, v, dout))
    assert ops.xna_backward_select(q5, k5, v5, ksz) == "rows"
    dq, dk, dv = ops.xna_backward(q5, k5, v5, g5, ksz)
    back = lambda t5: t5.permute(0, 1, 4, 2, 3).reshape(t5.shape[0], -1, *t5.shape[2:4]).float().cpu()
    # P and dS pass through bf16 (relative 2^-8) before the contractions over up to d^2 k^2 (query, key) pairs
    for got, ref, name in ((back(dq), rq, "dq"), (back(dk), rk, "dk"), (back(dv), rv, "dv")):
        scale = float(ref.abs().max())
        err = (got - ref).abs()
        assert float(err.max()) <= 2e-2 * scale + 1e-3 and float(err.mean()) <= 3e-3 * scale + 1e-4, \
            f"{name}: max err {float(err.max()):.3e} mean {float(err.mean()):.3e} (ref max {scale:.3e})"
    # the scalar table-driven kernel on the same inputs (fp32 throughout) brackets the oracle from the other side
    dq2, dk2, dv2 = ops.xna_backward(q5, k5, v5, g5, ksz, path="generic")
    for got, other, name in ((dk, dk2, "dk"), (dv, dv2, "dv")):
        scale = float(other.abs().max())
        assert float((got - other).abs().max()) <= 2e-2 * scale + 1e-3, name


@pytest.mark.parametrize("B,Cq,C,heads,lr,out_sz,ksz", [
    (4, 256, 768, 4, (16, 16), (32, 32), 9),     # the reference's OWN training step (train.py:113-133, config/base.yaml): ratio 2, Dv = 192
    (1, 256, 384, 4, (12, 12), (132, 132), 11),  # patch-11-like ratio 11 (no row tiles for the cell kernel: 5 of 16 lanes would idle), window 11, Dv = 96
    (1, 256, 384, 4, (12, 12), (168, 168), 11),  # ratio 14 with an 11 x 11 window: the cell kernel's partial row tiles stop at 9 x 9 (round 6), so this stays here
    (2, 256, 128, 4, (9, 11), (27, 44), 7),      # ratio (3, 4), Dv = 32 through the fragment path, ragged key tiles
    (1, 256, 1024, 4, (15, 15), (60, 60), 15),   # 15 x 15 window as tall as the grid, ratio 4, Dv = 256
    (1, 128, 6, 2, (10, 20), (20, 40), 5),       # ratio 2 with three value channels per head (gather form)
    (1, 256, 512, 4, (8, 8), (64, 64), 3),       # ratio 8, Dv = 128
])
def test_xna_backward_rows_integer_ratios(dev, B, Cq, C, heads, lr, out_sz, ksz):
    _rows_backward_integer_doc(dev, B, Cq, C, heads, lr, out_sz, ksz)


@pytest.mark.parametrize("B,Cq,C,heads,lr,out_sz,ksz", [
    (1, 256, 24, 4, (5, 7), (23, 30), 3),        # F4 shapes: irregular neighbourhoods, repeated taps, six value channels per head
    (4, 256, 768, 4, (13, 13), (32, 32), 9),     # `down_factor: random` training (utils/training.py:38-45): 512^2 * 0.4 -> 208^2 -> 13^2 features
    (1, 256, 384, 4, (28, 28), (64, 64), 9),     # the notebook's geometry (F9): ratio 16/7
    (1, 96, 3, 1, (20, 24), (30, 31), 7),        # one wide head, three value channels, ratio 1.5 / 1.29
    (2, 128, 64, 2, (9, 10), (31, 17), 5),       # ratio 3.4 / 1.7, Dv = 32
])
def test_xna_backward_rows_noninteger_ratios(dev, B, Cq, C, heads, lr, out_sz, ksz):
    """Non-integer ratios: taps repeat, so a (query, key) pair carries the product of its row and column multiplicities and the
    keys' inverse neighbourhoods are found by scanning the tables."""
    _rows_backward_case(dev, B, Cq, C, heads, lr, out_sz, ksz)


def _rows_backward_integer_doc(dev, B, Cq, C, heads, lr, out_sz, ksz):
    """Integer ratios the cell kernel does not take (cells narrower than a 16-pixel row tile, 15 x 15 windows): the row-streaming
    matrix-core backward with the key-stationary pass walking every query column chunk of the keys' inverse neighbourhood."""
    _rows_backward_case(dev, B, Cq, C, heads, lr, out_sz, ksz)


def test_rows_backward_keeps_a_planted_inf_inside_its_chunk_rows(dev):
    """A non-finite value stays inside the tiles whose streamed rows hold it (naf_hip.h, naf_xna_bwd).  Before C ABI 0.4.0 a chunk that
    skipped its upper 16 slots multiplied the STALE LDS rows of the wave's previous tile by zero weights, so one Inf could turn
    gradients of an unrelated tile -- another image of the batch -- into NaN.  Here: the denoising shape class (ratio 1, one head,
    window 7) on 40-pixel rows (tiles 0 and 1 of a row use their chunk's upper half, tile 2 skips it) with more tiles than the launch
    has waves (5 x 700 x 3 = 10 500 against 8 192), so that waves walk from a tile of image 0 to a tile of images 3 / 4."""
    from naf_amd import ops
    Bn, Cq, C, heads, size, ksz = 5, 64, 3, 1, (700, 40), 7
    r = ksz // 2
    q = bf16r(O.hash_normal((Bn, Cq, *size), 551))
    k = bf16r(O.hash_normal((Bn, Cq, *size), 552))
    v = bf16r(O.hash_normal((Bn, C, *size), 553))
    g = bf16r(O.hash_normal((Bn, C, *size), 554))
    q5, k5, v5, g5 = (to5(t, heads).to(dev) for t in (q, k, v, g))
    assert ops.xna_backward_select(q5, k5, v5, ksz) == "rows"
    clean = [t.float().cpu() for t in ops.xna_backward(q5, k5, v5, g5, ksz)]
    assert all(bool(torch.isfinite(t).all()) for t in clean)
    for which in ("v", "k", "q", "g"):
        t5 = {"v": v5, "k": k5, "q": q5, "g": g5}
        planted = {n: t.clone() for n, t in t5.items()}
        for y0 in range(20, 700, 40):                                   # many rows of image 0, every column tile
            planted[which][0, 0, y0, :, 1] = float("inf")
        got = [t.float().cpu() for t in ops.xna_backward(planted["q"], planted["k"], planted["v"], planted["g"], ksz)]
        far = torch.ones(size[0], dtype=torch.bool)
        for y0 in range(20, 700, 40):
            far[y0 - 2 * r: y0 + 2 * r + 1] = False
        for name, a, b in zip(("dq", "dk", "dv"), got, clean):
            assert torch.equal(a[1:], b[1:]), f"{which} = Inf in image 0 changed {name} of another image"
            assert torch.equal(a[0][:, far], b[0][:, far]), f"{which} = Inf changed rows of {name} that are two window radii away"
        assert not all(bool(torch.isfinite(t[0]).all()) for t in got)       # ... and it does surface where it belongs


def test_rows_backward_fuzz_against_table_driven_kernel(dev):
    """Seeded random geometries (ratios 1 .. 9 per axis, integer and not, every window size, ragged widths, both value forms): the
    row-streaming matrix-core backward against the independent scalar table-driven kernel (fp32 throughout) on the same bf16 inputs."""
    from naf_amd import ops
    rng = np.random.RandomState(4321)
    done = nonint = wide = 0
    for _ in range(600):
        ksz = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
        h, w = int(rng.randint(ksz, 30)), int(rng.randint(ksz, 30))
        if rng.rand() < 0.5:
            Ho, Wo = h * int(rng.randint(1, 5)), w * int(rng.randint(1, 5))
        else:
            Ho, Wo = max(h, int(h * rng.uniform(1.0, 9.0) ** rng.uniform(0.0, 1.0))), max(w, int(w * rng.uniform(1.0, 9.0) ** rng.uniform(0.0, 1.0)))
        if Ho * Wo > 90 * 90 or ksz * (Ho // h) > Ho or ksz * (Wo // w) > Wo:
            continue
        heads = int(rng.choice([1, 2]))
        if rng.rand() < 0.5:
            Dq, Dv = 64, int(rng.choice([32, 64, 96, 128, 192, 256]))
        else:
            Dq, Dv = int(rng.choice([64, 96, 128, 192, 256, 384, 512])), int(rng.randint(1, 33))
        q = (torch.randn(1, heads, Ho, Wo, Dq, device=dev) * (64.0 / Dq) ** 0.25).to(torch.bfloat16)
        k = (torch.randn(1, heads, h, w, Dq, device=dev) * (64.0 / Dq) ** 0.25).to(torch.bfloat16)
        v = torch.randn(1, h, w, heads, Dv, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        g = torch.randn(1, Ho, Wo, heads, Dv, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        if ops.xna_backward_select(q, k, v, ksz) != "rows":
            continue
        a = ops.xna_backward(q, k, v, g, ksz)
        b = ops.xna_backward(q, k, v, g, ksz, path="generic")
        for x, y, name in zip(a, b, ("dq", "dk", "dv")):
            scale = float(y.float().abs().max())
            err = float((x.float() - y.float()).abs().max())
            assert bool(torch.isfinite(x.float()).all()) and err <= 2.5e-2 * scale + 1e-3, (name, h, w, Ho, Wo, ksz, Dq, Dv, heads, err, scale)
        done += 1
        nonint += int(Ho % h != 0 or Wo % w != 0)
        wide += int(Dq > 64)
        if done >= 60:
            break
    assert done >= 40 and nonint >= 10 and wide >= 10, (done, nonint, wide)


def test_xna_autograd_function(dev):
    """ops.XnaFunction: torch autograd drives naf_xna_fwd / naf_xna_bwd."""
    from naf_amd import ops
    heads, ksz = 4, 5
    k = bf16r(O.hash_normal((1, 256, 6, 6), 512))
    q = bf16r(O.hash_normal((1, 256, 96, 96), 511))
    v = bf16r(O.hash_normal((1, 128, 6, 6), 513))
    q5, k5, v5 = (to5(t, heads).to(dev).requires_grad_(True) for t in (q, k, v))
    out = ops.XnaFunction.apply(q5, k5, v5, ksz, None, torch.float32)
    w = to5(O.hash_normal((1, 128, 96, 96), 514), heads).to(dev).float()
    (out * w).sum().backward()
    rq, rk, rv = O.xna_backward(q, k, v, bf16r(O.hash_normal((1, 128, 96, 96), 514)), ksz, heads)
    back = lambda t5: t5.permute(0, 1, 4, 2, 3).reshape(t5.shape[0], -1, *t5.shape[2:4]).float().cpu()
    for got, ref, name in ((back(q5.grad), rq, "dq"), (back(k5.grad), rk, "dk"), (back(v5.grad), rv, "dv")):
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 3e-2 * scale + 1e-3, name


def test_golden_F8_gradients_forward_train(dev, golden_dir):
    """NAF.forward_train gradients against the REFERENCE's own autograd (golden F8, generated by importing it)."""
    g = np.load(os.path.join(golden_dir, "F8_gradients.npz"))
    p = O.make_params(seed=int(g["param_seed"]))
    m = _load_model(dev, p, kernel_size=int(g["k"]))
    for prm in m.parameters():
        prm.requires_grad_(True)
    img = O.hash_normal(tuple(g["shape"]), int(g["image_seed"])).to(dev)
    ft = O.hash_normal(tuple(g["feat_shape"]), int(g["feat_seed"])).to(dev).requires_grad_(True)
    w = O.hash_normal((1, 128, 48, 48), int(g["weight_seed"])).to(dev)
    (m.forward_train(img, ft, (48, 48), amp=False).float() * w).sum().backward()      # fp32 torch stem: the reference's own precision
    ref = torch.from_numpy(g["dfeatures"])
    assert float((ft.grad.float().cpu() - ref).abs().max()) <= 3e-2 * float(ref.abs().max()) + 1e-3
    named = dict(m.named_parameters())
    for i, name in enumerate(g["names"]):
        ref = torch.from_numpy(g[f"g{i}"])
        got = named[str(name)].grad.float().cpu()
        if got.dim() == 4 and got.shape[1] == 128:
            got = got[::4, ::4]
        assert float((got - ref).abs().max()) <= 5e-2 * float(ref.abs().max()) + 1e-3, name


def test_forward_train_gradients_match_oracle(dev):
    """NAF.forward_train: loss gradients w.r.t. encoder parameters and features vs autograd through the oracle."""
    p = O.make_params(seed=21)
    m = _load_model(dev, p, kernel_size=3)
    img = O.hash_normal((1, 3, 48, 48), 521)
    ft = O.hash_normal((1, 128, 3, 3), 522)
    wgt = O.hash_normal((1, 128, 48, 48), 523)
    # oracle: fp32 autograd
    po = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "periods" not in k) for k, v in p.items()}
    fo = ft.clone().requires_grad_(True)
    (O.naf_forward(po, img, fo, (48, 48), kernel_size=3) * wgt).sum().backward()
    # HIP attention fwd + bwd, torch stem
    for prm in m.parameters():
        prm.requires_grad_(True)
    fd = ft.to(dev).requires_grad_(True)
    out = m.forward_train(img.to(dev), fd, (48, 48), amp=False)
    (out.float() * wgt.to(dev)).sum().backward()
    ref_out = O.naf_forward(p, img, ft, (48, 48), kernel_size=3)
    assert_close(out.float().cpu(), ref_out, 2e-2, 1e-2, "forward_train output")
    checked = 0
    for name, prm in m.named_parameters():
        ref = po[name].grad
        if ref is None:
            continue
        got = prm.grad.float().cpu()
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err <= 5e-2 * scale + 1e-3, f"{name}: grad err {err:.3e} vs max {scale:.3e}"
        checked += 1
    assert checked >= 20
    gs = float(fo.grad.abs().max())
    assert float((fd.grad.float().cpu() - fo.grad).abs().max()) <= 3e-2 * gs + 1e-3


def test_forward_train_amp_tracks_the_fp32_path(dev):
    """forward_train(amp=True) = the reference's use_bf16 mode (train.py:120): bf16 stem convolutions under autocast.
    Output and gradients stay within bf16-level distance of the fp32 training path."""
    p = O.make_params(seed=23)
    img = O.hash_normal((1, 3, 64, 64), 531).to(dev)
    ft0 = O.hash_normal((1, 128, 4, 4), 532).to(dev)
    wgt = O.hash_normal((1, 128, 64, 64), 533).to(dev)
    grads, outs = [], []
    for amp in (False, True):
        m = _load_model(dev, p, kernel_size=3)
        for prm in m.parameters():
            prm.requires_grad_(True)
        fd = ft0.clone().requires_grad_(True)
        out = m.forward_train(img, fd, (64, 64), amp=amp)
        (out.float() * wgt).sum().backward()
        outs.append(out.float())
        grads.append({n: q.grad.float() for n, q in m.named_parameters() if q.grad is not None} | {"features": fd.grad.float()})
    assert_close(outs[1].cpu(), outs[0].cpu(), 8e-2, 4e-2, "amp output")
    for n, g0 in grads[0].items():
        scale = float(g0.abs().max())
        err = float((grads[1][n] - g0).abs().max())
        assert err <= 0.12 * scale + 1e-3, f"{n}: amp grad err {err:.3e} vs max {scale:.3e}"


@pytest.mark.parametrize("shape,fmt", [((1, 3, 40, 56), "f32"), ((2, 3, 33, 47), "bf16_strided")])
def test_first_1x1_layer_recomputes_conv0_bit_exactly(dev, shape, fmt):
    """naf_stem_conv_fwd(first = conv0 args) == naf_stem_conv0_fwd + naf_stem_conv_fwd, bit for bit, and the
    statistics-only conv0 pass produces the same sums as the storing pass."""
    from naf_amd import ops
    B, _, H, W = shape
    img = O.hash_normal(shape, 601).to(dev)
    if fmt == "bf16_strided":
        img = img.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)      # NHWC storage
    w0 = (O.hash_normal((128, 3, 1, 1), 602) * 0.5).to(dev)
    b0 = (O.hash_normal((128,), 603) * 0.1).to(dev)
    gw = (1.0 + 0.1 * O.hash_normal((128,), 604)).to(dev)
    gb = (0.1 * O.hash_normal((128,), 605)).to(dev)
    wp = (O.hash_normal((1, 128, 128), 606) * 0.09).to(torch.bfloat16).to(dev)
    cb = (O.hash_normal((128,), 607) * 0.1).to(dev)
    x0 = torch.empty((B, H, W, 128), dtype=torch.bfloat16, device=dev)
    st = ops.new_stats(B, dev, lead=(4,))
    ops.stem_conv0(img, w0, b0, x0, st[0])
    ops.stem_conv0(img, w0, b0, None, st[1])        # statistics only: from the image's moments (fp64), no matrix work
    t0, t1 = ops.stats_total(st[0]), ops.stats_total(st[1])
    assert torch.allclose(t0, t1, rtol=2e-6, atol=1e-3), float((t0 - t1).abs().max())
    assert float(st[1][1:].abs().max()) == 0.0      # the moments' copies are cleared: copy 0 holds the sums
    ya, yb = torch.empty_like(x0), torch.empty_like(x0)
    ops.stem_conv(x0, st[0], gw, gb, 1e-5, wp, cb, ya, st[2])
    ops.stem_conv(None, st[0], gw, gb, 1e-5, wp, cb, yb, st[3], first=(img, w0, b0))     # same statistics: same bits
    assert torch.equal(ya, yb), float((ya.float() - yb.float()).abs().max())
    assert torch.allclose(ops.stats_total(st[2]), ops.stats_total(st[3]), rtol=1e-12, atol=1e-9)
    yc = torch.empty_like(x0)
    ops.stem_conv(None, st[1], gw, gb, 1e-5, wp, cb, yc, torch.zeros_like(st[3]), first=(img, w0, b0))
    assert float((ya.float() - yc.float()).abs().max()) <= 2.0 ** -6     # moments-based statistics: at most a bf16 ulp apart


def test_heads_rope_differs_from_heads_attn(dev):
    p = O.make_params(dim=64, heads_rope=1, seed=8)
    m = _load_model(dev, p, dim=64, heads_attn=4, heads_rope=1, kernel_size=3)
    img = O.hash_normal((1, 3, 16, 16), 81)
    ft = O.hash_normal((1, 8, 4, 4), 82)
    ref = O.naf_forward(p, img, ft, (16, 16), kernel_size=3, heads_attn=4, heads_rope=1)
    assert_close(m(img.to(dev), ft.to(dev), (16, 16)).float().cpu(), ref, 2e-2, 1e-2, "heads 1/4")


# ---- BASELINE full sizes: size-independent properties + sampled rows vs the oracle ----------------------
FULL = [
    ("G1", 1, 768, 64, 1024, 7, "f32"),
    ("G2-k7", 1, 1024, 32, 512, 7, "f32"),
    ("G2-k11", 1, 1024, 32, 512, 11, "f32"),
    ("G2-k15", 1, 1024, 32, 512, 15, "f32"),
    ("G4", 1, 768, 128, 2048, 7, "f32"),
    # BASELINE.json configs[3]: the 8-images-per-GPU shard of the B = 64, C = 1024 batch, in the dtype it runs in (bf16 out:
    # 17 GB per result instead of 34 GB)
    ("G3", 8, 1024, 64, 1024, 7, "bf16"),
]


@pytest.mark.parametrize("name,B,C,lr,out_sz,ksz,odt", FULL)
def test_full_size_properties(dev, name, B, C, lr, out_sz, ksz, odt):
    from naf_amd import ops
    heads = 4
    out_dtype = torch.float32 if odt == "f32" else torch.bfloat16
    # bf16 results carry one more rounding (2^-9 relative, values up to ~4)
    tol_lin_max, tol_lin_mean, tol_rows = (6e-2, 4e-3, 6e-3) if odt == "f32" else (1.2e-1, 8e-3, 1.2e-2)
    gen = torch.Generator(device="cpu").manual_seed(1234)
    k = torch.randn(B, 256, lr, lr, generator=gen)
    v1 = torch.randn(B, C, lr, lr, generator=gen)
    v2 = torch.randn(B, C, lr, lr, generator=gen)
    q5 = torch.randn(B, heads, out_sz, out_sz, 64, generator=torch.Generator(device=dev).manual_seed(5), device=dev,
                     dtype=torch.float32).to(torch.bfloat16)
    k5 = to5(k, heads).to(dev)

    def run(v):
        vp = ops.pack_values(v.to(dev))
        v5 = vp.view(B, lr, lr, heads, C // heads).permute(0, 3, 1, 2, 4)
        assert ops.xna_select(q5, k5, v5, ksz, out_dtype=out_dtype) == "mfma"
        return ops.xna_forward(q5, k5, v5, ksz, out_dtype=out_dtype, path="mfma")

    # (1) partition of unity: constant values pass through
    ones = run(torch.full((B, C, lr, lr), 0.5))
    assert float((ones.float() - 0.5).abs().max()) <= (2e-3 if odt == "f32" else 4e-3)
    del ones
    # (2) linearity in V: f(v1) + f(v2) == f(v1 + v2) up to bf16 rounding of the packed values
    o1, o2 = run(bf16r(v1)), run(bf16r(v2))
    o12 = run(bf16r(bf16r(v1) + bf16r(v2)))
    lmax, lsum = 0.0, 0.0
    for b in range(B):                                   # per image: bounds the fp32 temporaries at B = 8
        lin = (o1[b].float() + o2[b].float() - o12[b].float()).abs()
        lmax, lsum = max(lmax, float(lin.max())), lsum + float(lin.double().sum())
        del lin
    assert lmax <= tol_lin_max and lsum / o1.numel() <= tol_lin_mean, (lmax, lsum / o1.numel())
    # (3) convexity: outputs stay inside [min, max] of the values
    slack = 1e-3 if odt == "f32" else 3.2e-2
    assert float(o1.max()) <= float(bf16r(v1).max()) + slack and float(o1.min()) >= float(bf16r(v1).min()) - slack
    del o2, o12
    # (4) sampled rows against the oracle at full size (interior, top border, bottom border), every image of the batch
    rows = sorted({0, 1, out_sz // 2 - 1, out_sz // 2, out_sz - 17, out_sz - 1})
    iy = O.axis_index_table(out_sz, lr, ksz)[rows]
    ix = O.axis_index_table(out_sz, lr, ksz)
    q_rows = q5[:, :, rows].float().cpu().permute(0, 1, 4, 2, 3).reshape(B, 256, len(rows), out_sz)
    ref = O.xna_tables(q_rows, bf16r(k), bf16r(v1), iy, ix, heads)
    got = o1[:, :, rows].permute(0, 1, 4, 2, 3).reshape(B, C, len(rows), out_sz).float().cpu()
    assert_close(got, ref, tol_rows, tol_rows, f"{name} sampled rows")


BENCHED = [   # name, C, lr, out, window: the workloads bench.py times (BASELINE.json configs[1], [2], [4]), one image
    ("G1", 768, 64, 1024, 7),
    ("G2-k7", 1024, 32, 512, 7),
    ("G2-k11", 1024, 32, 512, 11),
    ("G2-k15", 1024, 32, 512, 15),
    ("G4", 768, 128, 2048, 7),
]


@pytest.mark.parametrize("name,C,lr,out_sz,ksz", BENCHED)
def test_full_size_benched_instantiations_bf16_rotate_on_load(dev, name, C, lr, out_sz, ksz):
    """VERDICT r02 (missing 3): the attention exactly as bench.py's forward runs it -- bf16 output, un-rotated channels-last
    guidance rotated on load, keys from the keys-only RoPE/pool pass -- at the benched sizes, i.e. the instantiations
    profiles/r0x_pmc_hbm_traffic.txt names (staged 8-wave / 4-wave cell kernels at k = 7, the bf16 sliding-window kernel at
    k = 11 / 15; G4's output is 3.2 G elements: offsets past 2^31), against the oracle's rope + pool + attention on sampled
    rows.  test_full_size_properties runs fp32 output and materialised queries, i.e. other kernels at these sizes."""
    from naf_amd import ops
    heads, Dq, B = 4, 64, 1
    gen = torch.Generator(device=dev).manual_seed(77)
    xd = torch.randn((B, out_sz, out_sz, heads * Dq), device=dev, generator=gen).to(torch.bfloat16).permute(0, 3, 1, 2)   # channels-last
    v = bf16r(torch.randn((B, C, lr, lr), generator=torch.Generator().manual_seed(78)))
    per = O.rope_periods(heads * Dq, heads, 100.0)
    ty, tx = ops.rope_tables(per.to(dev), out_sz, out_sz)
    none_q, k5 = ops.rope_pool(xd, ty, tx, heads, (lr, lr), write_q=False)
    assert none_q is None
    q_raw = xd.permute(0, 2, 3, 1).unflatten(3, (heads, Dq)).permute(0, 3, 1, 2, 4)
    assert ops.xna_rope_fusable(q_raw, (lr, lr), C // heads, ksz, (ty, tx), out_dtype=torch.bfloat16)
    vp = ops.pack_values(v.to(dev))
    v5 = vp.view(B, lr, lr, heads, C // heads).permute(0, 3, 1, 2, 4)
    out = ops.xna_forward(q_raw, k5, v5, ksz, out_dtype=torch.bfloat16, path="mfma", rope_tables=(ty, tx))
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (B, heads, out_sz, out_sz, C // heads)
    # oracle: rope head by head (bounds the fp32 temporaries at 2048^2), keys = pool(rope(x)), attention on sampled rows
    x = xd.float().cpu()
    d = out_sz // lr
    rows = sorted({0, 1, d - 1, d, out_sz // 2 - 1, out_sz // 2, out_sz - d - 1, out_sz - 2, out_sz - 1})
    xr_rows, keys = [], []
    for hd in range(heads):
        xr = O.rope(x[:, hd * Dq:(hd + 1) * Dq], per, 1)
        keys.append(O.key_pool(xr, (lr, lr)))
        xr_rows.append(xr[:, :, rows].clone())
        del xr
    ref_k = torch.cat(keys, dim=1)
    got_k = k5.permute(0, 1, 4, 2, 3).reshape(B, heads * Dq, lr, lr).float().cpu()
    assert_close(got_k, ref_k, 4e-3, 8e-3, f"{name} keys (bf16 of a {d}x{d} box mean)")
    q_rows = bf16r(torch.cat(xr_rows, dim=1))
    iy = O.axis_index_table(out_sz, lr, ksz)[rows]
    ix = O.axis_index_table(out_sz, lr, ksz)
    ref = O.xna_tables(q_rows, bf16r(ref_k), v, iy, ix, heads)
    got = out[:, :, rows].permute(0, 1, 4, 2, 3).reshape(B, C, len(rows), out_sz).float().cpu()
    assert_close(got, ref, 1.2e-2, 1.2e-2, f"{name} bf16 rotate-on-load, sampled rows")
    # partition of unity on the same kernel: constant values pass through every pixel of the output
    ones = ops.pack_values(torch.full((B, C, lr, lr), 0.5).to(dev))
    o1 = ops.xna_forward(q_raw, k5, ones.view(B, lr, lr, heads, C // heads).permute(0, 3, 1, 2, 4), ksz, out_dtype=torch.bfloat16,
                         path="mfma", rope_tables=(ty, tx))
    assert float((o1.float() - 0.5).abs().max()) <= 4e-3


def test_rccl_single_rank_sharded_forward(dev):
    """naf_amd.dist on the real RCCL backend ("nccl" on ROCm) with a one-rank group: parameter broadcast, batch
    sharding, result gathering and the sharded forward run through the same calls the multi-GPU bench uses."""
    import torch.distributed as dist
    from naf_amd import dist as nd
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    port = 29500 + (os.getpid() % 400)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        p = O.make_params(seed=3)
        m = _load_model(dev, p, kernel_size=3)
        nd.broadcast_parameters(m, src=0)
        sm = nd.ShardedNAF(m)
        img = O.hash_normal((2, 3, 32, 32), 701).to(dev)
        ft = O.hash_normal((2, 64, 4, 4), 702).to(dev)
        out = sm(img, ft, (32, 32))
        ref = m(img, ft, (32, 32))
        assert torch.equal(out, ref)
        lo, hi = nd.shard_range(2, 0, 1)
        mine = nd.scatter_batch(img, img.shape, img.dtype, dev, src=0)
        assert (lo, hi) == (0, 2) and torch.equal(mine, img)
        vals = nd.gather_scalars([1.5, 2.5], dev)
        assert vals == [[1.5, 2.5]]
        full = nd.gather_outputs(out, 2)
        assert torch.equal(full, ref)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("hw,lr,C,ksz", [((64, 64), (4, 4), 128, 3),      # rotate-on-load shape
                                          ((64, 64), (28, 28), 128, 9),    # non-integer ratio: tables built on the device, inside the graph
                                          ((42, 56), (3, 4), 64, 3)])      # 14x14 cells: materialised queries
def test_captured_forward_replays_bit_exactly(dev, hw, lr, C, ksz):
    """NAF.capture: the hipGraph replay gives the eager result, also after new inputs are copied in."""
    p = O.make_params(seed=5)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((1, 3, *hw), 801).to(dev)
    ft = O.hash_normal((1, C, *lr), 802).to(dev)
    g = m.capture(img, ft, hw)
    assert torch.equal(g(), m(img, ft, hw))
    img2 = O.hash_normal((1, 3, *hw), 803).to(dev)
    ft2 = O.hash_normal((1, C, *lr), 804).to(dev)
    assert torch.equal(g(img2, ft2), m(img2, ft2, hw))


def test_capture_from_a_train_mode_module_is_the_inference_forward(dev):
    """``hubconf.naf()`` returns a train-mode module (like the reference); capturing from it must record the
    fused inference forward -- not ``forward_train`` (torch stem with saved activations, a RoPE jitter draw frozen into the
    graph, an output with a grad_fn) -- and replays must equal the eval-mode forward bit for bit."""
    p = O.make_params(seed=5)
    m = _load_model(dev, p, kernel_size=3)
    img = O.hash_normal((1, 3, 64, 64), 811).to(dev)
    ft = O.hash_normal((1, 128, 4, 4), 812).to(dev)
    want = m(img, ft, (64, 64))
    m.train()
    assert any(q.requires_grad for q in m.parameters()) and torch.is_grad_enabled()
    g = m.capture(img, ft, (64, 64))
    out = g()
    assert out.grad_fn is None and not out.requires_grad
    assert torch.equal(out, want)
    assert torch.equal(g(), want)                                     # no per-replay randomness in the graph


def test_generic_conv0_statistics_and_null_check(dev):
    """the general-width first convolution (stem_generic.hip) now reduces its GroupNorm sums across the wave before
    the LDS atomics and guards the publish on ``stats_out`` -- which the C ABI requires for this entry anyway: a call without
    it is rejected by naf_stem_conv0_fwd's validation (an error string, no launch), never a device fault."""
    import torch.nn.functional as F
    from naf_amd import ops
    B, H, W, Cc = 1, 19, 23, 48
    img = O.hash_normal((B, 3, H, W), 821).to(dev)
    for ks in (1, 3):
        w = O.hash_normal((Cc, 3, ks, ks), 822 + ks).to(dev).contiguous()
        b = O.hash_normal((Cc,), 824).to(dev)
        y = torch.empty((B, H, W, Cc), dtype=torch.bfloat16, device=dev)
        st = ops.new_stats(B, dev)
        with pytest.raises(ValueError, match="NULL pointer"):
            ops.stem_conv0(img, w, b, y, None)
        ops.stem_conv0(img, w, b, y, st)
        torch.cuda.synchronize()
        x = F.pad(img, (1, 1, 1, 1), mode="reflect") if ks == 3 else img
        r = F.conv2d(x.double(), w.double(), b.double()).float().cpu()        # [B, Cc, H, W]
        assert_close(y.float().permute(0, 3, 1, 2).cpu(), r, 1e-5, 2.0 ** -7, f"conv0 k{ks}")
        rs = torch.stack([r.double().view(B, 8, -1).sum(-1), (r.double() ** 2).view(B, 8, -1).sum(-1)], dim=-1)
        assert torch.allclose(ops.stats_total(st).cpu(), rs, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("img_hw,lr,C,ksz", [
    ((50, 70), (5, 7), 32, 3),     # ratio 10: integer, but tiles straddle rows (generic tile loop, queries materialised)
    ((50, 70), (6, 9), 24, 3),     # non-integer ratio: table-driven attention
    ((96, 64), (6, 4), 64, 3),     # ratio 16: row tiles + rotate-on-load, odd strip / segment geometry in the stem
    ((33, 47), (33, 47), 16, 7),   # ratio 1 (denoising-like), image sizes that are multiples of nothing
])
def test_full_forward_odd_shapes_match_oracle(dev, img_hw, lr, C, ksz):
    """Whole NAF forward (fused stem, RoPE/pool, attention) on awkward geometries vs the fp32 oracle."""
    p = O.make_params(seed=31)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((1, 3, *img_hw), 901)
    ft = O.hash_normal((1, C, *lr), 902)
    ref = O.naf_forward(p, img, ft, img_hw, kernel_size=ksz)
    out = m(img.to(dev), ft.to(dev), img_hw).float().cpu()
    assert out.shape == ref.shape
    err = (out - ref).abs()
    assert float(err.max()) <= 2e-2 + 1e-2 * float(ref.abs().max()) and float(err.mean()) <= 6e-3, \
        f"max {float(err.max()):.3e} mean {float(err.mean()):.3e}"


@pytest.mark.parametrize("B,C,lr,out_sz,ksz,out_dtype", [
    (1, 192, (9, 9), (144, 144), 9, torch.float32),        # window = whole grid: the ring never slides
    (2, 256, (12, 20), (192, 320), 7, torch.float32),      # fp32 output -> sliding kernel, several segments per row
    (1, 192, (11, 13), (176, 416), 11, torch.bfloat16),    # 11x11, dx = 32 (two row tiles per cell row), Dv = 48 (odd tile count)
    (1, 1024, (15, 17), (240, 272), 15, torch.bfloat16),   # 15x15, Dv = 256: 155 KB window, 8 waves
    (1, 1024, (12, 13), (192, 208), 11, torch.bfloat16),   # 11x11, Dv = 256: stores staged 128 channels at a time (round 4)
    (2, 1024, (13, 14), (208, 448), 13, torch.bfloat16),   # 13x13, Dv = 256: staged 64 channels at a time, dx = 32, two images
])
def test_sliding_window_kernel_matches_oracle(dev, B, C, lr, out_sz, ksz, out_dtype):
    """xna_slide_kernel (plans without staged stores): ring-of-columns window, per-cell column refresh, segment borders."""
    heads = 4
    q = bf16r(O.hash_normal((B, 256, *out_sz), 951))
    k = bf16r(O.hash_normal((B, 256, *lr), 952))
    v = bf16r(O.hash_normal((B, C, *lr), 953))
    ref = O.xna_lowres(q, k, v, ksz, heads)
    out = run_xna(dev, q, k, v, ksz, heads, out_dtype=out_dtype, path="mfma")
    tol = 6e-3 if out_dtype == torch.float32 else 1.2e-2
    assert_close(out, ref, tol, tol, f"slide k={ksz}")


@pytest.mark.parametrize("feat_dtype", [torch.bfloat16, torch.float32])
def test_single_call_forward_equals_composed_calls(dev, feat_dtype):
    """naf_forward (the whole forward in one foreign call) == the same kernels launched one by one: bit for bit where both run the same
    kernels (8 x 8 pixel cells: keys from the pooling pass in both); on 16 x 16 pixel cells the one call takes its keys from the last
    stem layers (naf_stem_conv_keys_fwd) while the composed path runs the pooling pass -- the keys then differ by at most one bf16
    rounding, the outputs by one bf16 step on a handful of values (cf. test_forward_with_and_without_key_fusion_agree)."""
    p = O.make_params(seed=41)
    m = _load_model(dev, p, kernel_size=5)
    img = O.hash_normal((2, 3, 96, 128), 961).to(dev)
    ft8 = O.hash_normal((2, 128, 12, 16), 963).to(dev).to(feat_dtype)
    assert m._forward_plan(img, ft8, (96, 128)) is not None
    a8 = m(img, ft8, (96, 128))
    m.single_call = False
    b8 = m(img, ft8, (96, 128))
    m.single_call = True
    assert a8.dtype == b8.dtype and torch.equal(a8, b8)
    ft = O.hash_normal((2, 128, 6, 8), 962).to(dev).to(feat_dtype)
    assert m._forward_plan(img, ft, (96, 128)) is not None
    a = m(img, ft, (96, 128))
    m.single_call = False
    b = m(img, ft, (96, 128))
    m.single_call = True
    d = (a.float() - b.float()).abs()
    assert a.dtype == b.dtype and float(d.max()) <= 4e-3 and float(d.mean()) <= 1e-6, (float(d.max()), float(d.mean()))
    ref = O.naf_forward(p, img.cpu(), ft.float().cpu(), (96, 128), kernel_size=5)
    assert_close(a.float().cpu(), ref, 2e-2, 1e-2, "single-call forward vs oracle")
    # a different architecture (other window per axis) falls back to the composed path
    m.upsampler.kernel_size = (5, 3)
    assert m._forward_plan(img, ft, (96, 128)) is None


def test_single_call_forward_phase_events(dev):
    """naf_forward_args.phase_events (round 3): events recorded at the phase boundaries of the ONE call are in stream order, their
    phases add up to the call, the result is the same with and without them, and NULL entries are skipped."""
    from naf_amd import ops
    p = O.make_params(seed=44)
    m = _load_model(dev, p, kernel_size=7)
    img = O.hash_normal((1, 3, 256, 256), 971).to(dev)
    ft = O.hash_normal((1, 128, 16, 16), 972).to(dev).to(torch.bfloat16)
    plan = m._forward_plan(img, ft, (256, 256))
    assert plan is not None
    plan.streams = 2                                    # the branches side by side on the stream this host lends (ops.forward_aux)
    assert plan.planned_streams() == 2
    base = plan.run(img, ft).clone()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    out = plan.run(img, ft, phase_events=ev)
    torch.cuda.synchronize()
    assert torch.equal(out, base)
    # order of the entries on the caller's stream (version >= 200: the branches' block layers run on two streams): [0] start,
    # [1] first convolutions, [2] before / [7] after one 3x3 launch, [4] stem end (the second stream has joined), [5] attention
    # start, [6] attention end; [3] is recorded on the SECOND stream behind one 1x1 launch: after the fork [1], before the join [4]
    order = [0, 1, 2, 7, 4, 5, 6]
    gaps = [ev[order[i]].elapsed_time(ev[order[i + 1]]) for i in range(6)]
    assert all(g >= 0.0 for g in gaps), gaps
    total = ev[0].elapsed_time(ev[6])
    assert total > 0.0 and abs(sum(gaps) - total) <= 1e-3 + 0.05 * total
    assert gaps[2] > 0.0 and gaps[5] > 0.0                        # a 3x3 launch was bracketed; the attention ran
    assert ev[1].elapsed_time(ev[3]) > 0.0 and ev[3].elapsed_time(ev[4]) >= 0.0
    sparse = [None, None, ev[2], None, ev[4]]           # only some boundaries asked for
    assert torch.equal(plan.run(img, ft, phase_events=sparse), base)
    torch.cuda.synchronize()
    assert ev[2].elapsed_time(ev[4]) > 0.0
    # one stream (naf_forward, or NAF_FWD_ONE_STREAM): the layers alternate, [2] .. [3] brackets one 1x1 launch, [3] .. [7] one 3x3 launch
    plan.streams = 1
    assert plan.planned_streams() == 1
    out1 = plan.run(img, ft, phase_events=ev)
    torch.cuda.synchronize()
    assert torch.equal(out1, base)                      # same kernels, same bits
    order1 = [0, 1, 2, 3, 7, 4, 5, 6]
    gaps1 = [ev[order1[i]].elapsed_time(ev[order1[i + 1]]) for i in range(7)]
    assert all(g >= 0.0 for g in gaps1) and gaps1[2] > 0.0 and gaps1[3] > 0.0, gaps1


def test_forward_stream_plan_and_flags(dev):
    """naf_forward_streams / naf_forward_ex flags (C ABI 0.4.0): the library's plan is two streams except where the 3x3 layer launch
    is one full round of long-lived workgroups (512^2 at batch 1: profiles/r05_streams_rule.txt); either layout can be forced; the
    3x3 first convolution with exact products (NAF_FWD_CONV0_EXACT) differs from the default by bf16 roundings of 16-bit products."""
    p = O.make_params(seed=46)
    m = _load_model(dev, p, kernel_size=7)
    ft = O.hash_normal((1, 128, 32, 32), 992).to(dev).to(torch.bfloat16)
    for S, want in ((256, 2), (448, 2), (512, 1), (640, 2), (1024, 2)):
        img = O.hash_normal((1, 3, S, S), 991).to(dev)
        f = ft if S % 32 == 0 else ft[:, :, :S // 16, :S // 16]
        plan = m._forward_plan(img, f, (S, S))
        assert plan is not None and plan.planned_streams() == want, (S, plan.planned_streams())
    img = O.hash_normal((1, 3, 512, 512), 991).to(dev)
    plan = m._forward_plan(img, ft, (512, 512))
    outs = {}
    for st in (0, 1, 2):
        plan.streams = st
        assert plan.planned_streams() == (1 if st in (0, 1) else 2)
        outs[st] = plan.run(img, ft).clone()
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    plan.streams = 0
    plan.conv0_exact = True
    exact = plan.run(img, ft).clone()
    plan.conv0_exact = False
    torch.cuda.synchronize()
    d = (exact.float() - outs[0].float()).abs()
    assert 0.0 < float(d.max()) <= 3e-2 and float(d.mean()) <= 2e-4, (float(d.max()), float(d.mean()))
    # C ABI 0.4.2: a one-stream call takes the workspace WITHOUT the fourth activation buffer; a call that may fork does not
    import ctypes as C
    from naf_amd import _lib, ops
    assert plan.ws_bytes_one == plan.ws_bytes - ((512 * 512 * 128 * 2 + 255) // 256) * 256 and plan.planned_streams() == 1
    plan.release_workspaces()
    again = plan.run(img, ft)                                             # the library's plan here is one stream: the host allocates the smaller one
    assert plan._ws.numel() == plan.ws_bytes_one and torch.equal(again, outs[0])
    plan.streams = 2
    small = torch.empty((plan.ws_bytes_one,), dtype=torch.uint8, device=dev)
    a = plan.args
    a.workspace, a.workspace_bytes = small.data_ptr(), small.numel()
    with ops.AUX_POOL.lease(dev.index or 0, int(torch.cuda.current_stream(dev).cuda_stream)) as aux:
        rc = plan.lib.naf_forward_ex(C.byref(a), C.byref(aux), _lib.FWD_TWO_STREAMS, torch.cuda.current_stream(dev).cuda_stream)
    assert rc != 0 and b"workspace" in plan.lib.naf_last_error()
    plan.streams = 0
    ref = O.naf_forward(p, img.cpu(), ft.float().cpu(), (512, 512), kernel_size=7)
    assert_close(exact.float().cpu(), ref, 2e-2, 1e-2, "forward with exact first-convolution products vs oracle")
    assert_close(outs[0].float().cpu(), ref, 2e-2, 1e-2, "forward with 16-bit first-convolution products vs oracle")


def test_forward_error_after_the_fork_joins_the_lent_stream(dev):
    """VERDICT r04 / ADVICE r04: every return of naf_forward_ex behind the fork joins the second stream back into the caller's.
    feat_dtype = 7 passes the up-front validation and fails in the value packing -- the first launch on the lent stream.  Eagerly
    the next forward on the same stream and workspace is unharmed; under a hipGraph capture the failed call leaves no un-joined
    branch behind (0.3.x: hipErrorStreamCaptureUnjoined at the end of the capture)."""
    p = O.make_params(seed=47)
    m = _load_model(dev, p, kernel_size=7)
    img = O.hash_normal((1, 3, 256, 256), 995).to(dev)
    ft = O.hash_normal((1, 128, 16, 16), 996).to(dev).to(torch.bfloat16)
    plan = m._forward_plan(img, ft, (256, 256))
    plan.streams = 2
    base = plan.run(img, ft).clone()
    torch.cuda.synchronize()
    good = plan.args.feat_dtype
    for _ in range(3):
        plan.args.feat_dtype = 7
        with pytest.raises(ValueError, match="naf_pack_values"):
            plan.run(img, ft)
        plan.args.feat_dtype = good
        assert torch.equal(plan.run(img, ft), base)
    torch.cuda.synchronize()
    # the same failure inside a capture: the capture still ends cleanly and the graph (one good forward) replays
    side = torch.cuda.Stream(device=dev)
    from naf_amd import ops
    ops.forward_aux(dev, side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        plan.args.feat_dtype = 7
        with pytest.raises(ValueError):
            plan.run(img, ft)
        plan.args.feat_dtype = good
        out = plan.run(img, ft)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, base)


def test_two_host_threads_one_of_them_capturing(dev):
    """the library owns no stream any more -- every host thread lends its own (ops.forward_aux is keyed by
    thread, device and caller stream).  Thread A captures a forward into a hipGraph and replays it while thread B issues eager
    forwards of another module on another stream; neither sees the other's fork / join events or second stream."""
    import threading
    p = O.make_params(seed=48)
    ma, mb = _load_model(dev, p, kernel_size=7), _load_model(dev, p, kernel_size=7)
    img_a, img_b = O.hash_normal((1, 3, 256, 256), 997).to(dev), O.hash_normal((1, 3, 320, 256), 998).to(dev)
    ft_a = O.hash_normal((1, 128, 16, 16), 999).to(dev).to(torch.bfloat16)
    ft_b = O.hash_normal((1, 128, 20, 16), 1000).to(dev).to(torch.bfloat16)
    with torch.no_grad():
        ref_a, ref_b = ma(img_a, ft_a, (256, 256)).clone(), mb(img_b, ft_b, (320, 256)).clone()
    assert ma._forward_plan(img_a, ft_a, (256, 256)).planned_streams() == 2 and mb._forward_plan(img_b, ft_b, (320, 256)).planned_streams() == 2
    torch.cuda.synchronize()
    res, errs = {}, []
    start = threading.Barrier(2)

    def thread_a():
        try:
            torch.cuda.set_device(dev)
            start.wait(timeout=60)
            outs = []
            for _ in range(3):
                gf = ma.capture(img_a, ft_a, (256, 256), capture_error_mode="thread_local")
                outs.append(gf().clone())
                outs.append(gf().clone())
            res["a"] = outs
        except Exception as e:      # noqa: BLE001
            errs.append(("a", repr(e)))

    def thread_b():
        try:
            torch.cuda.set_device(dev)
            sb = torch.cuda.Stream(device=dev)
            start.wait(timeout=60)
            outs = []
            with torch.no_grad(), torch.cuda.stream(sb):
                for _ in range(40):
                    outs.append(mb(img_b, ft_b, (320, 256)))
            sb.synchronize()
            res["b"] = outs
        except Exception as e:      # noqa: BLE001
            errs.append(("b", repr(e)))

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(); tb.start()
    ta.join(timeout=300); tb.join(timeout=300)
    assert not ta.is_alive() and not tb.is_alive(), "a thread hung"
    assert not errs, errs
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref_a) for o in res["a"]) and all(torch.equal(o, ref_b) for o in res["b"])


@pytest.mark.parametrize("B,heads,lr,out_sz,Dq,Dv,ksz,kernel", [
    (2, 4, (16, 16), (32, 32), 64, 192, 9, "rows"),       # train.py's own step geometry: row-streaming kernel, tables + workspace
    (1, 4, (6, 5), (96, 80), 64, 32, 5, "mfma"),          # cell kernel: nothing but the tensors
    (1, 2, (7, 9), (7, 9), 80, 3, 3, "generic"),          # no matrix-core instantiation: table-driven scalar kernel, tables only
])
def test_c_host_backward_program_matches_python(dev, tmp_path, B, heads, lr, out_sz, Dq, Dv, ksz, kernel):
    """examples/c_host_bwd.c -- a plain C program (gcc, HIP runtime API, include/naf_hip.h) -- runs naf_xna_bwd on the same inputs as
    the Python host: it asks naf_xna_bwd_supported which kernel serves the shapes and brings the index tables / the statistics
    workspace that kernel wants.  dq matches bit for bit; dk / dv too unless partial sums meet by atomics (row sharing)."""
    import shutil, subprocess
    from naf_amd import _lib, ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / ROCm headers on this box")
    exe = str(tmp_path / "c_host_bwd")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gcc, "-O2", "-D__HIP_PLATFORM_AMD__", f"-I{root}/include", "-I/opt/rocm/include", f"{root}/examples/c_host_bwd.c", "-o", exe,
                    f"-L{libdir}", "-lnaf_hip", "-L/opt/rocm/lib", "-lamdhip64"], check=True, capture_output=True)
    mk = lambda shape, seed: O.hash_normal(shape, seed).to(torch.bfloat16)
    q = mk((B, *out_sz, heads, Dq), 1501); k = mk((B, *lr, heads, Dq), 1502)
    v = mk((B, *lr, heads, Dv), 1503); g = mk((B, *out_sz, heads, Dv), 1504)
    for t, name in ((q, "q"), (k, "k"), (v, "v"), (g, "dout")):
        (tmp_path / f"{name}.bin").write_bytes(t.contiguous().view(torch.int16).numpy().astype("<i2").tobytes())
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe] + [str(tmp_path / f"{n}.bin") for n in ("q", "k", "v", "dout", "dq", "dk", "dv")] +
                       [str(x) for x in (B, heads, *out_sz, *lr, Dq, Dv, ksz)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"kernel {kernel}" in r.stdout, r.stdout
    to5 = lambda t: t.to(dev).permute(0, 3, 1, 2, 4)
    dq, dk, dv = ops.xna_backward(to5(q), to5(k), to5(v), to5(g), ksz)
    assert ops.xna_backward_select(to5(q), to5(k), to5(v), ksz) == kernel
    got_dq = torch.from_numpy(np.frombuffer((tmp_path / "dq.bin").read_bytes(), dtype="<i2").copy()).view(torch.bfloat16).view(B, *out_sz, heads, Dq)
    got_dk = torch.from_numpy(np.frombuffer((tmp_path / "dk.bin").read_bytes(), dtype="<f4").copy()).view(B, *lr, heads, Dq)
    got_dv = torch.from_numpy(np.frombuffer((tmp_path / "dv.bin").read_bytes(), dtype="<f4").copy()).view(B, *lr, heads, Dv)
    assert torch.equal(got_dq, dq.permute(0, 2, 3, 1, 4).cpu().contiguous())
    for got, ref, name in ((got_dk, dk, "dk"), (got_dv, dv, "dv")):
        ref = ref.permute(0, 2, 3, 1, 4).cpu().contiguous()
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 1e-5 * scale + 1e-7, name      # fp32 atomics: the order of the partial sums is free


def test_forward_on_two_streams_does_not_share_scratch(dev):
    """a ForwardPlan used to own ONE workspace, so forwards of one module enqueued on two streams raced on the
    stem's activation buffers.  Every (device, stream) now gets its own: two different inputs enqueued back to back on two streams give
    exactly what they give one after the other."""
    p = O.make_params(seed=45)
    m = _load_model(dev, p, kernel_size=7)
    img_a, img_b = O.hash_normal((1, 3, 512, 512), 981).to(dev), O.hash_normal((1, 3, 512, 512), 982).to(dev)
    ft = O.hash_normal((1, 128, 32, 32), 983).to(dev).to(torch.bfloat16)
    ref_a, ref_b = m(img_a, ft, (512, 512)).clone(), m(img_b, ft, (512, 512)).clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for _ in range(3):
        with torch.cuda.stream(s1):
            out_a = m(img_a, ft, (512, 512))
        with torch.cuda.stream(s2):
            out_b = m(img_b, ft, (512, 512))
        torch.cuda.synchronize()
        assert torch.equal(out_a, ref_a) and torch.equal(out_b, ref_b)


def test_single_call_forward_alternating_order_equals_sequential(dev):
    """The stem's launch order (the two branches' layers alternate, three rotating activation buffers) does not change a bit of the
    result: the same forward through a library instance with NAF_STEM_ORDER=0 (one branch after the other) in a child process."""
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from naf_amd import NAF\n"
        "torch.manual_seed(5)\n"
        "m = NAF(kernel_size=5).cuda().eval()\n"
        "img = torch.rand(2, 3, 96, 80).cuda(); ft = torch.randn(2, 64, 6, 5).cuda()\n"
        "with torch.no_grad(): o = m(img, ft, (96, 80))\n"
        "torch.save(o.float().cpu(), sys.argv[1])\n")
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for order in ("0", None):
            env = dict(os.environ, NAF_HIP_KNOBS="1")
            env.pop("NAF_STEM_ORDER", None)
            if order is not None:
                env["NAF_STEM_ORDER"] = order
            f = os.path.join(td, f"o{order}.pt")
            r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(f))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("hw,lr,C,ksz,path", [
    ((70, 84), (5, 6), 128, 5, "mfma"),        # 14x14 cells: cell kernel, queries materialised (tiles straddle rows)
    ((64, 64), (28, 28), 128, 9, "union"),     # non-integer ratio: index tables built on the device
    ((40, 48), (20, 24), 256, 7, "union"),     # 2x2 cells
    ((45, 45), (45, 45), 64, 3, "union"),      # ratio 1
    ((23, 30), (5, 7), 24, 3, "rows"),         # Dv = 6: row-streaming kernel
])
def test_single_call_forward_other_geometries(dev, hw, lr, C, ksz, path):
    """naf_forward beyond the rotate-on-load shapes: same kernels, same bits as the composed path; the index tables it
    builds on the device are those of the host function."""
    from naf_amd import ops
    p = O.make_params(seed=43)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((1, 3, *hw), 971).to(dev)
    ft = O.hash_normal((1, C, *lr), 972).to(dev)
    assert m._forward_plan(img, ft, hw) is not None
    a = m(img, ft, hw)
    class Names:                       # ops.KERNEL_TIMER protocol: which launches the composed path makes
        enabled = False

        def __init__(self):
            self.seen = set()

        def start(self, name):
            self.seen.add(name)

        def stop(self, name):
            pass

    m.single_call = False
    ops.KERNEL_TIMER = rec = Names()
    try:
        b = m(img, ft, hw)
    finally:
        ops.KERNEL_TIMER = None
    m.single_call = True
    assert ("xna_" + path) in rec.seen, sorted(rec.seen)
    assert torch.equal(a, b)
    ref = O.naf_forward(p, img.cpu(), ft.float().cpu(), hw, kernel_size=ksz)
    # cells of 1 or 2 pixels: each query's own cell dominates its window (q.k of the cell it was pooled from), the softmax is
    # peaked and the output follows single keys instead of averaging them -- the stem's bf16 error shows undamped
    # Measured (profiles/r04_tolerance_budget.txt): 2-pixel cells max 3.2e-2 with 2 of 491 520 elements outside 8c, 1-pixel cells
    # 4.1e-2 with 19 of 129 600; asserted: the maximum with 1.5x head-room and three times the outliers.
    cell = hw[0] // lr[0]
    got = a.float().cpu()
    if cell <= 2:
        assert_close(got, ref, 3.6e-2 if cell == 2 else 6e-2, 1e-2, f"single-call forward vs oracle {hw} {lr}")
        outside = ((got - ref).abs() > 2e-2 + 1e-2 * ref.abs()).float().mean()
        assert float(outside) <= (1.3e-5 if cell == 2 else 4.5e-4), float(outside)
    else:
        assert_close(got, ref, 2e-2, 1e-2, f"single-call forward vs oracle {hw} {lr}")


@pytest.mark.parametrize("L_out,L_in,k", [(64, 28, 9), (512, 37, 9), (518, 37, 15), (23, 5, 3), (30, 7, 5), (256, 256, 15),
                                          (1024, 64, 7), (333, 41, 7), (100, 99, 11)])
def test_device_index_table_equals_host_table(dev, L_out, L_in, k):
    import ctypes as C
    from naf_amd import ops, _lib
    host = ops.axis_index_table(L_out, L_in, k)
    d = torch.empty((L_out, k), dtype=torch.int32, device=dev)
    rc = _lib.load().naf_axis_index_table_device(C.cast(d.data_ptr(), C.POINTER(C.c_int32)), L_out, L_in, k,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert torch.equal(d.cpu(), host)


@pytest.mark.parametrize("shape,out", [((2, 64, 40, 56), (10, 14)), ((1, 256, 45, 70), (45, 23)), ((1, 8, 33, 47), (7, 40)), ((1, 256, 64, 64), (64, 64)),
                                       ((1, 64, 20, 24), (50, 31))])
def test_pool_guidance_matches_torch(dev, shape, out):
    """naf_pool_guidance == F.adaptive_avg_pool2d on the bf16 channels-last guidance (fp32 accumulation, one rounding)."""
    import torch.nn.functional as F
    from naf_amd import ops
    x = O.hash_normal(shape, 1201).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    got = ops.pool_guidance(x, out).float().cpu()
    ref = F.adaptive_avg_pool2d(x.float().cpu(), out)
    assert_close(got, ref, 1e-6, 2.0 ** -8, f"pool {shape} -> {out}")


@pytest.mark.parametrize("img_hw,out_hw,lr,C,ksz", [((96, 128), (48, 64), (6, 8), 128, 5),      # image 2x the output, 8x8 cells (union)
                                                    ((100, 90), (32, 32), (2, 2), 64, 1),       # 3.1x / 2.8x, 16x16 cells (rotate on load)
                                                    ((64, 64), (28, 28), (14, 14), 64, 3),      # pooled AND 2x2 cells
                                                    ((40, 48), (64, 64), (4, 4), 64, 3)])       # output LARGER than the image
def test_single_call_forward_with_pooled_guidance(dev, img_hw, out_hw, lr, C, ksz):
    """Image larger than the output (naf.py:34, the reference's 'out 56^2 ... 224^2 from image 448^2' rows): naf_forward
    pools the guidance itself; same bits as the composed path, same values as the oracle."""
    p = O.make_params(seed=45)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((1, 3, *img_hw), 981).to(dev)
    ft = O.hash_normal((1, C, *lr), 982).to(dev)
    assert m._forward_plan(img, ft, out_hw) is not None
    a = m(img, ft, out_hw)
    m.single_call = False
    b = m(img, ft, out_hw)
    m.single_call = True
    assert a.shape == (1, C, *out_hw) and torch.equal(a, b)
    ref = O.naf_forward(p, img.cpu(), ft.float().cpu(), out_hw, kernel_size=ksz)
    assert_close(a.float().cpu(), ref, 2e-2, 1e-2, f"pooled forward vs oracle {img_hw} -> {out_hw}")


@pytest.mark.parametrize("shape,size,fmt", [((2, 3, 97, 130), (24, 32), "f32"), ((1, 3, 64, 200), (40, 40), "bf16_nhwc"), ((1, 3, 33, 33), (32, 8), "f32")])
def test_preshrink_matches_torch_bilinear(dev, shape, size, fmt):
    """naf_preshrink_image == F.interpolate(mode="bilinear", align_corners=False) (naf.py:39-48), ATen's arithmetic."""
    import torch.nn.functional as F
    from naf_amd import ops
    img = O.hash_normal(shape, 1301)
    if fmt == "bf16_nhwc":
        img = img.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    got = ops.preshrink_image(img.to(dev), size).cpu()
    ref = F.interpolate(img.float(), size=size, mode="bilinear", align_corners=False)
    assert_close(got, ref, 2e-6, 2e-6, f"preshrink {shape} -> {size}")


@pytest.mark.parametrize("img_hw,out_hw,lr,C,ksz", [((200, 136), (32, 32), (2, 2), 64, 1),      # 6.25x / 4.25x: shrunk to 128 x 128, then pooled
                                                    ((130, 40), (16, 32), (4, 8), 128, 3)])     # only the height exceeds 4x
def test_single_call_forward_with_preshrunk_image(dev, img_hw, out_hw, lr, C, ksz):
    """Image more than 4x the output (naf.py:39-48): naf_forward shrinks it itself; same bits as the composed path."""
    p = O.make_params(seed=47)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((1, 3, *img_hw), 991).to(dev)
    ft = O.hash_normal((1, C, *lr), 992).to(dev)
    assert m._forward_plan(img, ft, out_hw) is not None
    a = m(img, ft, out_hw)
    m.single_call = False
    b = m(img, ft, out_hw)
    m.single_call = True
    assert a.shape == (1, C, *out_hw) and torch.equal(a, b)
    ref = O.naf_forward(p, img.cpu(), ft.float().cpu(), out_hw, kernel_size=ksz)
    assert_close(a.float().cpu(), ref, 2e-2, 1e-2, f"pre-shrunk forward vs oracle {img_hw} -> {out_hw}")


@pytest.mark.parametrize("img_hw,lr,C,ksz", [((64, 64), (4, 4), 128, 3),       # rotate-on-load cell kernel writes the logits
                                             ((64, 64), (28, 28), 64, 9),      # non-integer ratio: scalar table kernel with logits
                                             ((42, 56), (3, 4), 64, 3)])       # 14-pixel cells
def test_single_call_forward_return_weights(dev, img_hw, lr, C, ksz):
    """return_weights=True through naf_forward (logits field) == the composed path, and both match the oracle."""
    p = O.make_params(seed=49)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((1, 3, *img_hw), 995).to(dev)
    ft = O.hash_normal((1, C, *lr), 996).to(dev)
    a, wa = m(img, ft, img_hw, return_weights=True)
    m.single_call = False
    b, wb = m(img, ft, img_hw, return_weights=True)
    m.single_call = True
    assert torch.equal(a, b) and torch.equal(wa, wb) and wa.shape == (1, 4, *img_hw, ksz * ksz)
    ref, ref_w = O.naf_forward(p, img.cpu(), ft.float().cpu(), img_hw, kernel_size=ksz, return_weights=True)
    assert_close(a.float().cpu(), ref, 2e-2, 1e-2, "out with return_weights")
    assert float((wa.cpu() - ref_w).abs().mean()) <= 2e-2


@pytest.mark.parametrize("heads_rope,heads_attn", [(1, 4), (8, 2), (2, 4)])
def test_single_call_forward_with_different_rope_heads(dev, heads_rope, heads_attn):
    """heads_rope != heads_attn at the default width (naf.py:73-85): naf_forward rotates / pools with the RoPE head split and
    attends with the attention head split; same bits as the composed path, same values as the oracle."""
    p = O.make_params(dim=256, heads_rope=heads_rope, seed=51)
    m = _load_model(dev, p, dim=256, heads_attn=heads_attn, heads_rope=heads_rope, kernel_size=3)
    img = O.hash_normal((1, 3, 64, 64), 997).to(dev)
    ft = O.hash_normal((1, 64, 4, 4), 998).to(dev)
    assert m._forward_plan(img, ft, (64, 64)) is not None
    a = m(img, ft, (64, 64))
    m.single_call = False
    b = m(img, ft, (64, 64))
    m.single_call = True
    assert torch.equal(a, b)
    ref = O.naf_forward(p, img.cpu(), ft.float().cpu(), (64, 64), kernel_size=3, heads_attn=heads_attn, heads_rope=heads_rope)
    assert_close(a.float().cpu(), ref, 2e-2, 1e-2, f"heads_rope {heads_rope} heads_attn {heads_attn}")


@pytest.mark.parametrize("img_hw,out_hw,lr,C,ksz", [((64, 64), (64, 64), (4, 4), 128, 3), ((96, 80), (48, 40), (6, 5), 64, 5)])
def test_c_host_program_matches_python(dev, tmp_path, img_hw, out_hw, lr, C, ksz):
    """examples/c_host.c -- a plain C program (gcc, HIP runtime API, include/naf_hip.h, no Python / torch in the process)
    -- runs naf_forward on the same parameters and inputs and writes the same bits as the Python module."""
    import shutil, struct, subprocess
    from naf_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / ROCm headers on this box")
    exe = str(tmp_path / "c_host")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gcc, "-O2", "-D__HIP_PLATFORM_AMD__", f"-I{root}/include", "-I/opt/rocm/include", f"{root}/examples/c_host.c", "-o", exe,
                    f"-L{libdir}", "-lnaf_hip", "-L/opt/rocm/lib", "-lamdhip64"], check=True, capture_output=True)
    p = O.make_params(seed=53)
    m = _load_model(dev, p, kernel_size=ksz)
    enc = m.image_encoder
    blob = bytearray()
    periods = enc.rope.periods.detach().float().cpu().numpy()
    hdr = [4, enc.encoder[0].kernel_size[0], enc.encoder[1].conv1.kernel_size[0], enc.sem_encoder[0].kernel_size[0],
           enc.sem_encoder[1].conv1.kernel_size[0], len(periods), 4, 0]
    blob += struct.pack("<8i", *hdr) + periods.astype("<f4").tobytes()
    for seq in (enc.encoder, enc.sem_encoder):
        blob += seq[0].weight.detach().float().cpu().numpy().astype("<f4").tobytes() + seq[0].bias.detach().float().cpu().numpy().astype("<f4").tobytes()
        for blk in list(seq)[1:]:
            for norm, conv in ((blk.norm1, blk.conv1), (blk.norm2, blk.conv2)):
                blob += norm.weight.detach().float().cpu().numpy().astype("<f4").tobytes() + norm.bias.detach().float().cpu().numpy().astype("<f4").tobytes()
                blob += enc._packed(conv).cpu().view(torch.int16).numpy().astype("<i2").tobytes()
                blob += conv.bias.detach().float().cpu().numpy().astype("<f4").tobytes()
    (tmp_path / "params.bin").write_bytes(bytes(blob))
    img = O.hash_normal((2, 3, *img_hw), 1401)
    ft = O.hash_normal((2, C, *lr), 1402)
    (tmp_path / "image.bin").write_bytes(img.numpy().astype("<f4").tobytes())
    (tmp_path / "features.bin").write_bytes(ft.numpy().astype("<f4").tobytes())
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(tmp_path / "params.bin"), str(tmp_path / "image.bin"), str(tmp_path / "features.bin"), str(tmp_path / "out.bin"),
                        "2", str(img_hw[0]), str(img_hw[1]), str(lr[0]), str(lr[1]), str(C), str(ksz), str(out_hw[0]), str(out_hw[1])],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = torch.from_numpy(np.frombuffer((tmp_path / "out.bin").read_bytes(), dtype="<f4").copy()).view(2, *out_hw, C)
    ref = m(img.to(dev), ft.to(dev), out_hw)                       # Python host, same library
    assert ref.dtype == torch.float32 and torch.equal(got, ref.permute(0, 2, 3, 1).cpu().contiguous()), r.stdout


def test_forward_train_rope_augmentation(dev):
    """forward_train in .train() mode draws the reference's coordinate rescale (rope.py:117-122) per call; in .eval() mode
    its torch tables are the tables the HIP kernels use."""
    import naf_amd.model as M
    p = O.make_params(seed=55)
    m = _load_model(dev, p, kernel_size=3)
    ty, tx = m.image_encoder.rope.tables(48, 64)
    ey, ex = M._rope_train_tables(m.image_encoder.rope, 48, 64)
    assert float((ty - ey).abs().max()) <= 2e-6 and float((tx - ex).abs().max()) <= 2e-6
    img = O.hash_normal((1, 3, 64, 64), 1501).to(dev)
    ft = O.hash_normal((1, 128, 4, 4), 1502).to(dev)
    a = m.forward_train(img, ft, (64, 64), amp=False)
    a2 = m.forward_train(img, ft, (64, 64), amp=False)                                   # eval: same coordinates (MIOpen may pick another
    assert float((a.detach().float() - a2.detach().float()).abs().max()) <= 2e-2   # convolution algorithm from call to call)
    m.train()
    torch.manual_seed(0)
    b1 = m.forward_train(img, ft, (64, 64), amp=False)
    b2 = m.forward_train(img, ft, (64, 64), amp=False)
    assert torch.isfinite(b1).all()
    assert float((b1.detach().float() - b2.detach().float()).abs().mean()) > 1e-3 and float((a.detach().float() - b1.detach().float()).abs().mean()) > 1e-3   # a new rescale per call
    b1.float().sum().backward()                                                # and it is differentiable
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in m.image_encoder.parameters() if q.requires_grad)
