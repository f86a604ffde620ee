"""CPU tests of the oracle: (1) against the golden vectors produced by the imported reference
(oracle/make_golden.py), (2) NATTEN-independent identities that pin the neighbourhood restatement."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import naf_oracle as O

TOL = 1e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_hash_normal_is_stable():
    x = O.hash_normal((4, 5), seed=7)
    # known-answer: first values and a checksum, guards against platform drift of the generator
    assert x.dtype == torch.float32
    assert abs(float(x.double().sum()) - float(O.hash_normal((20,), seed=7).double().sum())) == 0.0
    big = O.hash_normal((100000,), seed=1)
    assert abs(float(big.mean())) < 0.02 and abs(float(big.std()) - 1.0) < 0.02
    assert np.array_equal(O.hash_normal((3,), 5).numpy(), O.hash_normal((3,), 5).numpy())


def test_golden_F1_rope(golden_dir):
    g = _load(golden_dir, "F1_rope")
    x = O.hash_normal(tuple(g["shape"]), int(g["seed"]))
    y = O.rope(x, torch.from_numpy(g["periods"]), int(g["heads"]))
    assert np.abs(y.numpy() - g["out"]).max() <= TOL
    assert np.array_equal(O.rope_periods(256, 4, 100.0).numpy(), g["periods"])


def test_golden_F2_conv_stem(golden_dir):
    g = _load(golden_dir, "F2_conv_stem")
    p = O.make_params(dim=int(g["dim"]), heads_rope=int(g["heads_rope"]), seed=int(g["param_seed"]))
    img = O.hash_normal(tuple(g["image_shape"]), int(g["image_seed"]))
    assert np.abs(O.conv_stem(img, p).numpy() - g["out"]).max() <= TOL


def test_golden_F3_xna(golden_dir):
    g = _load(golden_dir, "F3_xna_d4_k7")
    q = O.hash_normal(tuple(g["q_shape"]), int(g["q_seed"]))
    k = O.hash_normal((1, 256, *g["lr"]), int(g["k_seed"]))
    v = O.hash_normal((1, int(g["C"]), *g["lr"]), int(g["v_seed"]))
    out, lg = O.xna(q, k, v, int(g["k"]), int(g["heads"]), return_logits=True)
    assert np.abs(out.numpy() - g["out"]).max() <= TOL
    assert np.abs(lg.numpy() - g["logits"]).max() <= TOL
    assert np.abs(O.xna_lowres(q, k, v, int(g["k"]), int(g["heads"])).numpy() - g["out"]).max() <= TOL


def test_golden_F4_nonmultiple(golden_dir):
    g = _load(golden_dir, "F4_xna_nonmultiple")
    q = O.hash_normal(tuple(g["q_shape"]), int(g["q_seed"]))
    k = O.hash_normal((1, 128, *g["lr"]), int(g["k_seed"]))
    v = O.hash_normal((1, int(g["C"]), *g["lr"]), int(g["v_seed"]))
    for kk in (3, 5):
        out, lg = O.xna(q, k, v, kk, int(g["heads"]), return_logits=True)
        assert np.abs(out.numpy() - g[f"out_k{kk}"]).max() <= TOL
        assert np.abs(lg.numpy() - g[f"logits_k{kk}"]).max() <= TOL


def test_golden_F9_noninteger_ratio(golden_dir):
    """Notebook geometry 28^2 -> 64^2 and a rectangular 37^2 -> 100x150 case, whole forward, window 9."""
    g = _load(golden_dir, "F9_noninteger_ratio")
    p = O.make_params(seed=int(g["param_seed"]))
    for tag in ("a", "b"):
        H, W, h, w, C, iseed = (int(v) for v in g[f"{tag}_shape"])
        out = O.naf_forward(p, O.hash_normal((1, 3, H, W), iseed), O.hash_normal((1, C, h, w), iseed + 1), (H, W), kernel_size=int(g["k"]))
        got = out[:, :, 1::2, ::2] if tag == "a" else out[:, ::4, 1::2, ::3]
        assert np.abs(got.numpy() - g[f"{tag}_sample"]).max() <= TOL


def test_golden_F10_patch14(golden_dir):
    g = _load(golden_dir, "F10_patch14")
    p = O.make_params(seed=int(g["param_seed"]))
    H, W, h, w, C = (int(v) for v in g["shape"])
    out = O.naf_forward(p, O.hash_normal((1, 3, H, W), int(g["image_seed"])), O.hash_normal((1, C, h, w), int(g["feat_seed"])),
                        (H, W), kernel_size=int(g["k"]))
    assert np.abs(out[:, ::2, ::3, 1::3].numpy() - g["sample"]).max() <= TOL


def test_golden_F5_full_P1(golden_dir):
    """BASELINE configs[0]: 1x3x224x224 image, 1x384x14x14 features -> 224x224, window 7."""
    g = _load(golden_dir, "F5_full_P1")
    p = O.make_params(seed=int(g["param_seed"]))
    img = O.hash_normal((1, 3, 224, 224), int(g["image_seed"]))
    ft = O.hash_normal((1, 384, 14, 14), int(g["feat_seed"]))
    out = O.naf_forward_fast(p, img, ft, (224, 224), kernel_size=int(g["k"]))
    oy, ox = g["offset"]
    st = int(g["stride"])
    assert np.abs(out[:, :, oy::st, ox::st].numpy() - g["sample"]).max() <= TOL
    assert np.abs(out.mean(dim=(0, 2, 3)).numpy() - g["ch_mean"]).max() <= TOL
    assert np.abs(out.abs().amax(dim=(0, 2, 3)).numpy() - g["ch_absmax"]).max() <= TOL
    assert np.abs(out[:, ::48, :2, :].numpy() - g["top_rows"]).max() <= TOL
    assert np.abs(out[:, ::48, :, -2:].numpy() - g["left_cols"]).max() <= TOL


def test_golden_F6_denoise(golden_dir):
    g = _load(golden_dir, "F6_denoise_d1")
    p = O.make_params(dim=int(g["dim"]), heads_rope=1, seed=int(g["param_seed"]))
    shp = tuple(g["shape"])
    img, ft = O.hash_normal(shp, int(g["image_seed"])), O.hash_normal(shp, int(g["feat_seed"]))
    out, lg = O.naf_forward(p, img, ft, shp[-2:], kernel_size=int(g["k"]), heads_attn=1, heads_rope=1, return_weights=True)
    assert np.abs(out.numpy() - g["out"]).max() <= TOL
    assert np.abs(lg.numpy() - g["logits"]).max() <= TOL


def test_golden_F7_preshrink_pool(golden_dir):
    g = _load(golden_dir, "F7_preshrink_pool")
    p = O.make_params(dim=int(g["dim"]), heads_rope=2, seed=int(g["param_seed"]))
    ft = O.hash_normal(tuple(g["feat_shape"]), int(g["feat_seed"]))
    for tag in ("a", "b"):
        img = O.hash_normal(tuple(g[f"image_shape_{tag}"]), int(g[f"image_seed_{tag}"]))
        out = O.naf_forward(p, img, ft, tuple(g[f"out_size_{tag}"]), kernel_size=int(g["k"]), heads_attn=2, heads_rope=2)
        assert np.abs(out.numpy() - g[f"out_{tag}"]).max() <= TOL


# ---- NATTEN-independent identities -------------------------------------------
def test_full_window_equals_dense_attention():
    """k == h == w: every clamped window is the whole grid -> plain softmax(QK^T * scale) V."""
    h = w = 5
    d = 3
    q = O.hash_normal((2, 64, h * d, w * d), 11)
    k = O.hash_normal((2, 64, h, w), 12)
    v = O.hash_normal((2, 12, h, w), 13)
    out = O.xna(q, k, v, 5, 2)
    qh = q.reshape(2, 2, 32, -1).transpose(-1, -2)
    kh = k.reshape(2, 2, 32, -1).transpose(-1, -2)
    vh = v.reshape(2, 2, 6, -1).transpose(-1, -2)
    ref = F.scaled_dot_product_attention(qh, kh, vh)                   # default scale = D^-0.5
    ref = ref.transpose(-1, -2).reshape(2, 12, h * d, w * d)
    assert (out - ref).abs().max() <= 2e-6


def test_constant_values_give_constant_output():
    q = O.hash_normal((1, 32, 12, 18), 21)
    k = O.hash_normal((1, 32, 4, 6), 22)
    v = torch.full((1, 6, 4, 6), 0.75)
    out = O.xna(q, k, v, 3, 2)
    assert (out - 0.75).abs().max() <= 1e-6


def test_ratio_one_interior_equals_unfold():
    """d = 1: ordinary neighbourhood attention; interior pixels == an F.unfold computation."""
    H, W, kk, D, C = 9, 11, 3, 16, 4
    q = O.hash_normal((1, D, H, W), 31)
    k = O.hash_normal((1, D, H, W), 32)
    v = O.hash_normal((1, C, H, W), 33)
    out = O.xna(q, k, v, kk, 1)
    ku = F.unfold(k, kk, padding=1).reshape(1, D, kk * kk, H, W)
    vu = F.unfold(v, kk, padding=1).reshape(1, C, kk * kk, H, W)
    lg = torch.einsum("bdhw,bdkhw->bkhw", q, ku) * D ** -0.5
    ref = torch.einsum("bkhw,bckhw->bchw", lg.softmax(dim=1), vu)
    assert (out[..., 1:-1, 1:-1] - ref[..., 1:-1, 1:-1]).abs().max() <= 1e-6


@pytest.mark.parametrize("L_in,d,k", [(8, 4, 7), (14, 16, 7), (9, 2, 9), (7, 3, 7), (16, 8, 15), (5, 1, 3), (32, 16, 11)])
def test_lowres_form_equals_dilated_form(L_in, d, k):
    """Two independent derivations of the neighbourhood: NATTEN's dilated rule on the upsampled grid
    vs the clamped low-res window."""
    assert np.array_equal(O.axis_index_table(L_in * d, L_in, k), O.lowres_window_table(L_in * d, L_in, k))


def test_dilated_rule_equals_independent_strided_groups():
    """Fifth NATTEN-independent identity (round 6).  Dilation in neighbourhood attention is DEFINED (DiNAT, Hassani & Shi 2022: dilated
    neighbourhoods) as delta independent, undilated neighbourhood attentions on the strided sub-grids {m, m + delta, m + 2 delta, ...}: pixel
    i = m + delta j attends to the clamped k-window around j inside its own sub-grid of length ceil((L - m) / delta).  The restated
    ``get_window_start`` -- including its remainder branch for axis lengths that are not multiples of the dilation (the non-integer-ratio
    calls of the notebooks, golden F4 / F9) -- must be exactly that, for every axis length, window and dilation NATTEN accepts (k delta <= L)."""
    checked = 0
    for L in range(3, 90):
        for k in (3, 5, 7, 9, 11, 13, 15):
            for dil in range(1, 17):
                if k * dil > L:
                    continue
                for i in range(L):
                    m, j = i % dil, i // dil
                    n_group = (L - m + dil - 1) // dil                       # pixels of residue class m
                    start = min(max(j - k // 2, 0), n_group - k)             # plain clamped window inside the sub-grid
                    assert O.natten_window_start(i, L, k, dil) == m + dil * start, (L, k, dil, i)
                    checked += 1
    assert checked > 200000


def test_nearest_exact_matches_torch_off_ties():
    for L_in, L_out in [(5, 23), (7, 30), (14, 224), (28, 64), (28, 128), (6, 13), (3, 10)]:
        x = torch.arange(L_in, dtype=torch.float32).view(1, 1, 1, L_in)
        y = F.interpolate(x, size=(1, L_out), mode="nearest-exact").view(-1).long().numpy()
        assert np.array_equal(y, O.nearest_exact_src(L_out, L_in))


def test_natten_preconditions_raise():
    with pytest.raises(ValueError):
        O.axis_index_table(20, 5, 7)      # 7 * 4 > 20
    with pytest.raises(ValueError):
        O.axis_index_table(4, 8, 3)       # dilation 0


def test_xna_backward_matches_finite_differences():
    """oracle.xna_backward (autograd through the restated forward) against central differences of the forward."""
    torch.manual_seed(0)
    heads, ks = 2, 3
    q = torch.randn(1, 2 * 8, 8, 8, dtype=torch.float64)
    k = torch.randn(1, 2 * 8, 4, 4, dtype=torch.float64)
    v = torch.randn(1, 2 * 4, 4, 4, dtype=torch.float64)
    dout = torch.randn(1, 2 * 4, 8, 8, dtype=torch.float64)
    dq, dk, dv = O.xna_backward(q, k, v, dout, ks, heads)
    f = lambda qq, kk, vv: float((O.xna(qq, kk, vv, ks, heads) * dout).sum())
    eps = 1e-5
    gen = torch.Generator().manual_seed(1)
    for t, g, name in ((q, dq, "dq"), (k, dk, "dk"), (v, dv, "dv")):
        for _ in range(6):
            idx = tuple(int(torch.randint(0, n, (1,), generator=gen)) for n in t.shape)
            tp, tm = t.clone(), t.clone()
            tp[idx] += eps
            tm[idx] -= eps
            args_p = [tp if x is t else x for x in (q, k, v)]
            args_m = [tm if x is t else x for x in (q, k, v)]
            fd = (f(*args_p) - f(*args_m)) / (2 * eps)
            assert abs(fd - float(g[idx])) <= 1e-5 + 1e-4 * abs(fd), (name, idx, fd, float(g[idx]))


def _f8_sample(g, name_grad):
    return name_grad[::4, ::4] if name_grad.dim() == 4 and name_grad.shape[1] == 128 else name_grad


def test_golden_F8_gradients(golden_dir):
    """Autograd through the oracle == autograd through the imported reference (train-style loss), F8."""
    g = np.load(os.path.join(golden_dir, "F8_gradients.npz"))
    p = O.make_params(seed=int(g["param_seed"]))
    img = O.hash_normal(tuple(g["shape"]), int(g["image_seed"]))
    ft = O.hash_normal(tuple(g["feat_shape"]), int(g["feat_seed"])).requires_grad_(True)
    w = O.hash_normal((1, 128, 48, 48), int(g["weight_seed"]))
    po = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "periods" not in k) for k, v in p.items()}
    (O.naf_forward(po, img, ft, (48, 48), kernel_size=int(g["k"])) * w).sum().backward()
    ref = torch.from_numpy(g["dfeatures"])
    assert float((ft.grad - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    for i, name in enumerate(g["names"]):
        ref = torch.from_numpy(g[f"g{i}"])
        got = _f8_sample(g, po[str(name)].grad)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), name
