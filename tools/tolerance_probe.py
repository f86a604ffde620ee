"""Error budget of the whole-forward parity assertions: runs the tests whose tolerances are looser than
SURVEY 8c's 2e-2 + 1e-2 |ref| with `assert_close` replaced by a recorder and prints, per assertion, the measured maximum error,
the fraction of elements outside 2e-2 + 1e-2 |ref| and the tolerance the test states.  python tools/tolerance_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_parity as T
import test_gpu_fullsize as F

rows = []
def rec(got, ref, atol, rtol, what=""):
    err = (got - ref).abs()
    out = err > 2e-2 + 1e-2 * ref.abs()
    rows.append((what, float(err.max()), float(err.mean()), int(out.sum()), out.numel(), float((err - 1e-2 * ref.abs()).max()), atol, rtol,
                 float((err / (atol + rtol * ref.abs())).max())))
T.assert_close = rec
F._assert_close = rec
dev = torch.device("cuda:0")
gd = os.path.join(ROOT, "tests", "golden")
def run(name, fn):
    n0 = len(rows)
    try:
        fn()
    except Exception as e:      # later assertions of the test may not hold with a recorder in place of assert_close
        print(f"  ({name}: stopped at {type(e).__name__}: {str(e)[:80]})")
    for r in rows[n0:]:
        print(f"{name:34s} {r[0][:44]:44s} max {r[1]:.3e} mean {r[2]:.2e} outside-8c {r[3]:6d}/{r[4]:<8d} max(err - 1e-2|ref|) {r[5]:.3e}   stated {r[6]:g} + {r[7]:g}|ref| (used {r[8]:.2f})")
run("F5", lambda: T.test_golden_F5_full_forward_P1(dev, gd))
run("F10 f32", lambda: T.test_golden_F10_patch14(dev, gd, torch.float32))
run("F10 bf16", lambda: T.test_golden_F10_patch14(dev, gd, torch.bfloat16))
run("F6", lambda: T.test_golden_F6_denoise_like(dev, gd))
for hw, lr, C, ksz, path in (((40, 48), (20, 24), 256, 7, "union"), ((45, 45), (45, 45), 64, 3, "union")):
    run(f"cells of {hw[0] // lr[0]} px", lambda: T.test_single_call_forward_other_geometries(dev, hw, lr, C, ksz, path))
run("denoising cfg", lambda: F.test_denoising_configuration_runs_entirely_on_hip(dev))
run("forward_train", lambda: F.test_forward_is_differentiable_when_a_gradient_is_wanted(dev))
# smoke's two cases
import __graft_entry__ as G
from oracle import naf_oracle as O
from naf_amd import NAF
p = O.make_params(seed=11)
model = NAF(kernel_size=7).eval(); model.load_state_dict(p, strict=True); model = model.to(dev)
img = O.hash_normal((1, 3, 64, 64), 111); ft = O.hash_normal((1, 128, 8, 8), 112)
out = model(img.to(dev), ft.to(dev), (64, 64)); ref = O.naf_forward(p, img, ft, (64, 64), kernel_size=7)
run("smoke mfma", lambda: rec(out.float().cpu(), ref, 6e-2, 3e-2, "smoke 64x64"))
out2 = model(img.to(dev), ft.to(dev), (60, 52)); ref2 = O.naf_forward(p, img, ft, (60, 52), kernel_size=7)
run("smoke generic", lambda: rec(out2.float().cpu(), ref2, 6e-2, 3e-2, "smoke 60x52"))
