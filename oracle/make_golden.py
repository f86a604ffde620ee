"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON in the build container, and
check oracle/naf_oracle.py against it on the way.  TEST INFRASTRUCTURE; container-only.

    python oracle/make_golden.py            # needs /root/reference (read-only) on disk

The reference (valeoai/NAF, /root/reference) is imported unmodified; the only foreign piece is the
``natten.functional`` stand-in of oracle/natten_shim.py (NATTEN is not installable here ->
"parity unpinned" at that boundary, see naf_oracle.py's header).  Inputs and weights are NOT
stored: they are regenerated anywhere from ``naf_oracle.hash_normal`` seeds recorded in each file.
Only reference OUTPUTS (or strided samples of them) are stored, as float32 arrays.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NAF_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import natten_shim            # noqa: E402
from oracle import naf_oracle as O        # noqa: E402

natten_shim.install()
sys.path.insert(0, REF)
import importlib.util                      # noqa: E402

import src.layers as ref_layers           # noqa: E402  (reference: src/layers/__init__.py)

try:                                       # reference: src/model/naf.py (the package __init__ drags in
    from src.model.naf import NAF as RefNAF   # unrelated baselines that may need absent deps)
except Exception:                          # pragma: no cover
    spec = importlib.util.spec_from_file_location("ref_naf", os.path.join(REF, "src/model/naf.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    RefNAF = mod.NAF

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_grad_enabled(False)


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def ref_model(params, **kw):
    m = RefNAF(**kw).eval()
    missing = m.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


report = {}

# ---- F1: RoPE ---------------------------------------------------------------------------------
x = O.hash_normal((1, 256, 6, 10), seed=101)
rr = ref_layers.RoPE(embed_dim=256, num_heads=4, base=100.0, rescale_coords=2.0).eval()
y_ref = rr(x)
y_orc = O.rope(x, O.rope_periods(256, 4, 100.0), 4)
report["F1 rope"] = maxdiff(y_ref, y_orc)
assert maxdiff(rr.periods, O.rope_periods(256, 4, 100.0)) == 0.0
save("F1_rope", seed=101, shape=[1, 256, 6, 10], heads=4, out=y_ref, periods=rr.periods)

# ---- F2: conv stem (tiny dims; reflect pad, GN, SiLU, no residual) ------------------------------
p2 = O.make_params(dim=32, heads_rope=2, seed=2)
m2 = ref_model(p2, dim=32, heads_attn=2, heads_rope=2, kernel_size=3)
img2 = O.hash_normal((1, 3, 20, 24), seed=202)
e_ref = m2.image_encoder.forward_encoder(img2, (20, 24))
e_orc = O.conv_stem(img2, p2)
report["F2 conv stem"] = maxdiff(e_ref, e_orc)
save("F2_conv_stem", param_seed=2, dim=32, heads_rope=2, image_seed=202, image_shape=[1, 3, 20, 24], out=e_ref)

# ---- F3: attention only, integer ratio d=4, k=7, all border cells, with logits -------------------
q3 = O.hash_normal((1, 256, 32, 48), seed=301)
k3 = O.hash_normal((1, 256, 8, 12), seed=302)
v3 = O.hash_normal((1, 24, 8, 12), seed=303)
ca = ref_layers.CrossAttention(dim=256, num_heads=4, kernel_size=(7, 7))
o_ref, l_ref = ca(q3, k3, v3, return_weights=True)
o_orc, l_orc = O.xna(q3, k3, v3, 7, 4, return_logits=True)
report["F3 xna out"] = maxdiff(o_ref, o_orc)
report["F3 xna logits"] = maxdiff(l_ref, l_orc)
report["F3 lowres form"] = maxdiff(o_ref, O.xna_lowres(q3, k3, v3, 7, 4))
save("F3_xna_d4_k7", q_seed=301, k_seed=302, v_seed=303, q_shape=[1, 256, 32, 48], lr=[8, 12], C=24,
     heads=4, k=7, out=o_ref, logits=l_ref)

# ---- F4: non-multiple sizes 5x7 -> 23x30 (dilation 4,4 with remainders), k=3 and k=5 --------------
q4 = O.hash_normal((1, 128, 23, 30), seed=401)
k4 = O.hash_normal((1, 128, 5, 7), seed=402)
v4 = O.hash_normal((1, 16, 5, 7), seed=403)
f4 = {}
for kk in (3, 5):
    ca = ref_layers.CrossAttention(dim=128, num_heads=2, kernel_size=(kk, kk))
    o_ref, l_ref = ca(q4, k4, v4, return_weights=True)
    o_orc, l_orc = O.xna(q4, k4, v4, kk, 2, return_logits=True)
    report[f"F4 k={kk} out"] = maxdiff(o_ref, o_orc)
    report[f"F4 k={kk} logits"] = maxdiff(l_ref, l_orc)
    f4[f"out_k{kk}"] = o_ref
    f4[f"logits_k{kk}"] = l_ref
save("F4_xna_nonmultiple", q_seed=401, k_seed=402, v_seed=403, q_shape=[1, 128, 23, 30], lr=[5, 7], C=16,
     heads=2, **f4)

# ---- F5: full P1 config (BASELINE configs[0]) -- 224^2, C=384, 14^2 -> 224^2, k=7 ------------------
p5 = O.make_params(dim=256, heads_rope=4, seed=5)
m5 = ref_model(p5, kernel_size=7)
img5 = O.hash_normal((1, 3, 224, 224), seed=501)
ft5 = O.hash_normal((1, 384, 14, 14), seed=502)
o_ref = m5(img5, ft5, (224, 224))
o_orc = O.naf_forward(p5, img5, ft5, (224, 224), kernel_size=7)
o_fast = O.naf_forward_fast(p5, img5, ft5, (224, 224), kernel_size=7)
report["F5 full P1"] = maxdiff(o_ref, o_orc)
report["F5 full P1 (fast form)"] = maxdiff(o_ref, o_fast)
save("F5_full_P1", param_seed=5, image_seed=501, feat_seed=502, k=7, stride=16, offset=[3, 5],
     sample=o_ref[:, :, 3::16, 5::16].contiguous(), ch_mean=o_ref.mean(dim=(0, 2, 3)),
     ch_absmax=o_ref.abs().amax(dim=(0, 2, 3)), top_rows=o_ref[:, ::48, :2, :].contiguous(),
     left_cols=o_ref[:, ::48, :, -2:].contiguous())

# ---- F6: denoising-like call (ratio 1, C=3, one head; denoising.py:213) ---------------------------
p6 = O.make_params(dim=64, heads_rope=1, seed=6)
m6 = ref_model(p6, dim=64, heads_attn=1, heads_rope=1, kernel_size=5)
img6 = O.hash_normal((2, 3, 24, 20), seed=601)
ft6 = O.hash_normal((2, 3, 24, 20), seed=602)
o_ref, l_ref = m6(img6, ft6, (24, 20), return_weights=True)
o_orc, l_orc = O.naf_forward(p6, img6, ft6, (24, 20), kernel_size=5, heads_attn=1, heads_rope=1,
                             return_weights=True)
report["F6 denoise out"] = maxdiff(o_ref, o_orc)
report["F6 denoise logits"] = maxdiff(l_ref, l_orc)
save("F6_denoise_d1", param_seed=6, dim=64, image_seed=601, feat_seed=602, shape=[2, 3, 24, 20], k=5,
     out=o_ref, logits=l_ref)

# ---- F7: image > 4x output (bilinear pre-shrink, naf.py:39-48) and image != output pooling (:34) ----
p7 = O.make_params(dim=32, heads_rope=2, seed=7)
m7 = ref_model(p7, dim=32, heads_attn=2, heads_rope=2, kernel_size=3)
img7 = O.hash_normal((1, 3, 64, 80), seed=701)
ft7 = O.hash_normal((1, 10, 6, 7), seed=702)
o_ref_a = m7(img7, ft7, (12, 14))                                     # pre-shrink to 48x48, pool 48->12x14
o_orc_a = O.naf_forward(p7, img7, ft7, (12, 14), kernel_size=3, heads_attn=2, heads_rope=2)
img7b = O.hash_normal((1, 3, 36, 42), seed=703)
o_ref_b = m7(img7b, ft7, (18, 21))                                    # 2x pooling, no pre-shrink
o_orc_b = O.naf_forward(p7, img7b, ft7, (18, 21), kernel_size=3, heads_attn=2, heads_rope=2)
report["F7a preshrink"] = maxdiff(o_ref_a, o_orc_a)
report["F7b pooled"] = maxdiff(o_ref_b, o_orc_b)
save("F7_preshrink_pool", param_seed=7, dim=32, image_seed_a=701, image_shape_a=[1, 3, 64, 80], out_size_a=[12, 14],
     image_seed_b=703, image_shape_b=[1, 3, 36, 42], out_size_b=[18, 21], feat_seed=702,
     feat_shape=[1, 10, 6, 7], k=3, out_a=o_ref_a, out_b=o_ref_b)

# ---- F8: gradients (train.py:127-137): loss = sum(out * w) through the REFERENCE's own modules ----------
# d = 16 so that the HIP backward's MFMA kernel serves the shape (48 -> 3x3 cells, window 3); dim 256 (default width).
p8 = O.make_params(seed=8)
m8 = ref_model(p8, kernel_size=3)
img8 = O.hash_normal((1, 3, 48, 48), seed=801)
ft8 = O.hash_normal((1, 128, 3, 3), seed=802).requires_grad_(True)
w8 = O.hash_normal((1, 128, 48, 48), seed=803)
with torch.enable_grad():
    for prm in m8.parameters():
        prm.requires_grad_(True)
    (m8(img8, ft8, (48, 48)) * w8).sum().backward()
g_ref = {k_: v_.grad.detach().clone() for k_, v_ in m8.named_parameters() if v_.grad is not None}
g_ref_ft = ft8.grad.detach().clone()
with torch.enable_grad():
    po = {k_: v_.clone().requires_grad_(v_.dtype.is_floating_point and "periods" not in k_) for k_, v_ in p8.items()}
    fo = ft8.detach().clone().requires_grad_(True)
    (O.naf_forward(po, img8, fo, (48, 48), kernel_size=3) * w8).sum().backward()
worst8 = maxdiff(g_ref_ft, fo.grad) / max(1.0, float(g_ref_ft.abs().max()))
for k_, v_ in g_ref.items():
    worst8 = max(worst8, maxdiff(v_, po[k_].grad) / max(1.0, float(v_.abs().max())))
report["F8 gradients (rel. to max)"] = worst8
keep = ["image_encoder.encoder.0.weight", "image_encoder.encoder.2.conv2.weight", "image_encoder.encoder.1.norm1.weight",
        "image_encoder.sem_encoder.0.bias", "image_encoder.sem_encoder.1.conv1.weight", "image_encoder.sem_encoder.2.norm2.bias"]
save("F8_gradients", param_seed=8, image_seed=801, feat_seed=802, weight_seed=803, shape=[1, 3, 48, 48], feat_shape=[1, 128, 3, 3],
     k=3, dfeatures=g_ref_ft, names=np.array(keep),
     **{f"g{i}": (g_ref[n][::4, ::4] if g_ref[n].dim() == 4 and g_ref[n].shape[1] == 128 else g_ref[n]) for i, n in enumerate(keep)})
# (the two 128x128 conv-weight gradients are stored as [::4, ::4] samples: 32 x 32 x k x k)

# ---- F9: non-integer ratios end to end (notebooks/inference.ipynb cell 19: 28^2 -> 64^2; a rectangular 2.7 x 4.05 case):
#      nearest-exact K/V upsampling + dilation floor(ratio) inside the reference's CrossAttention, repeated taps ---------
p9 = O.make_params(dim=256, heads_rope=4, seed=9)
m9 = ref_model(p9, kernel_size=9)
f9 = {}
for tag, hw, lr9, C9, iseed in (("a", (64, 64), (28, 28), 64, 901), ("b", (100, 150), (37, 37), 128, 903)):
    img9 = O.hash_normal((1, 3, *hw), seed=iseed)
    ft9 = O.hash_normal((1, C9, *lr9), seed=iseed + 1)
    o_ref = m9(img9, ft9, hw)
    report[f"F9{tag} non-integer ratio"] = maxdiff(o_ref, O.naf_forward(p9, img9, ft9, hw, kernel_size=9))
    f9[f"{tag}_sample"] = (o_ref[:, :, 1::2, ::2] if tag == "a" else o_ref[:, ::4, 1::2, ::3]).contiguous()   # strided samples
    f9[f"{tag}_shape"] = np.array([*hw, *lr9, C9, iseed])
save("F9_noninteger_ratio", param_seed=9, k=9, **f9)

# ---- F10: patch-14 geometry end to end (DINOv2-style backbones, vit_wrapper.py:19-21: 14-pixel cells) -------------
p10 = O.make_params(dim=256, heads_rope=4, seed=10)
m10 = ref_model(p10, kernel_size=5)
img10 = O.hash_normal((1, 3, 140, 168), seed=1001)
ft10 = O.hash_normal((1, 64, 10, 12), seed=1002)
o_ref = m10(img10, ft10, (140, 168))
report["F10 patch-14 cells"] = maxdiff(o_ref, O.naf_forward(p10, img10, ft10, (140, 168), kernel_size=5))
save("F10_patch14", param_seed=10, image_seed=1001, feat_seed=1002, k=5, shape=[140, 168, 10, 12, 64],
     sample=o_ref[:, ::2, ::3, 1::3].contiguous())   # every 2nd channel, 3rd row, 3rd column

print("\noracle vs imported reference (max abs diff, fp32):")
worst = 0.0
for k_, v_ in report.items():
    print(f"  {k_:28s} {v_:.3e}")
    if not k_.startswith("F8"):
        worst = max(worst, v_)
assert worst <= 1e-5, f"oracle deviates from the reference by {worst}"
# gradients: fp32 autograd through two differently ordered (but mathematically identical) graphs
assert report["F8 gradients (rel. to max)"] <= 1e-4, report["F8 gradients (rel. to max)"]
with open(os.path.join(OUT, "REPORT.txt"), "w") as f:
    f.write("oracle/naf_oracle.py vs imported reference + natten shim (max abs diff, fp32)\n")
    for k_, v_ in report.items():
        f.write(f"{k_:28s} {v_:.3e}\n")
print("OK: oracle pinned to the imported reference (<= 1e-5) on F1-F10")
