"""GPU parity of the key pooling that rides on a branch's last stem layer (naf_stem_conv_keys_fwd): the layer's output must be
the bits naf_stem_conv_fwd writes, and the keys must be the oracle's pool(RoPE(guidance)) (naf.py:63-69 after rope.py:139-153)
of exactly those bf16 values -- to one bf16 rounding, the tolerance test_rope_pool holds the separate pre-pass to."""
import pytest
import torch

from oracle import naf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    from naf_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def bf16r(x):
    return x.to(torch.bfloat16).float()


def _layer_inputs(dev, B, H, W, ks, seed):
    x = bf16r(O.hash_normal((B, 128, H, W), seed) * 1.5 + 0.3)
    w = bf16r(O.hash_normal((128, 128, ks, ks), seed + 1, 1.0 / (11.3 * ks)))
    bias = O.hash_normal((128,), seed + 2, 0.1)
    gw, gb = 1.0 + O.hash_normal((128,), seed + 3, 0.1), O.hash_normal((128,), seed + 4, 0.1)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
    g = x.double().view(B, 8, 16, H, W)
    from naf_amd import ops
    st_in = ops.stats_from_total(torch.stack([g.sum(dim=(2, 3, 4)), (g * g).sum(dim=(2, 3, 4))], dim=-1).to(dev))
    wp = ops.pack_conv_weight(w).to(dev)
    return xd, st_in, gw.to(dev), gb.to(dev), wp, bias.to(dev)


@pytest.mark.parametrize("ks", [1, 3])
@pytest.mark.parametrize("shape", [(1, 32, 32), (2, 48, 64), (1, 128, 96), (3, 16, 160)])
@pytest.mark.parametrize("branch", [0, 1])
def test_stem_conv_keys_matches_the_separate_pass(dev, ks, shape, branch):
    from naf_amd import ops, _lib
    B, H, W = shape
    h, w = H // 16, W // 16
    xd, st_in, gw, gb, wp, bias = _layer_inputs(dev, B, H, W, ks, 900 + 10 * ks)
    per = O.rope_periods(256, 4, 100.0)
    ty, tx = ops.rope_tables(per.to(dev), H, W)
    # the layer alone, into its slice of the concatenated guidance
    cat0 = torch.zeros((B, H, W, 256), dtype=torch.bfloat16, device=dev)
    ops.stem_conv(xd, st_in, gw, gb, 1e-5, wp, bias, cat0[..., 128 * branch:128 * branch + 128], None)
    # the layer with the keys riding on it
    cat1 = torch.zeros((B, H, W, 256), dtype=torch.bfloat16, device=dev)
    keys = torch.full((B, h, w, 256), 7.0, dtype=torch.bfloat16, device=dev)
    ksl = keys[..., 128 * branch:128 * branch + 128]
    ops.stem_conv(xd, st_in, gw, gb, 1e-5, wp, bias, cat1[..., 128 * branch:128 * branch + 128], None, keys=(ksl, ty, tx))
    torch.cuda.synchronize()
    assert torch.equal(cat0, cat1), "the layer's output changed"
    other = keys[..., 128 * (1 - branch):128 * (1 - branch) + 128]
    assert bool((other == 7.0).all()), "wrote outside its key slice"
    # oracle: rotate the bf16 guidance this branch wrote (heads of 64 channels), pool to the cells
    y = cat1[..., 128 * branch:128 * branch + 128].float().cpu().permute(0, 3, 1, 2).contiguous()
    ref = O.key_pool(O.rope(y, per, 2), (h, w))
    got = ksl.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    assert bool((err <= 1e-5 + 2 ** -8 * ref.abs()).all()), f"keys ks={ks} {shape}: max err {float(err.max()):.3e}"


def test_stem_conv_keys_refuses_other_geometries(dev):
    from naf_amd import ops
    B, H, W = 1, 40, 48     # not 16-pixel cells
    xd, st_in, gw, gb, wp, bias = _layer_inputs(dev, B, H, W, 1, 950)
    per = O.rope_periods(256, 4, 100.0)
    ty, tx = ops.rope_tables(per.to(dev), H, W)
    y = torch.empty((B, H, W, 128), dtype=torch.bfloat16, device=dev)
    keys = torch.zeros((B, 2, 3, 128), dtype=torch.bfloat16, device=dev)
    with pytest.raises(Exception):
        ops.stem_conv(xd, st_in, gw, gb, 1e-5, wp, bias, y, None, keys=(keys, ty, tx))


def test_stem_conv_keys_fuzz_sizes(dev):
    """Seeded random geometries (cells of 16 x 16, strips of 32 for the 3x3 kernel, 1-3 images, 1-9 bands of cells): both kernels,
    keys vs the oracle on the layer's own bf16 output, as in the parametrised test above."""
    import numpy as np
    from naf_amd import ops
    rng = np.random.RandomState(404)
    per = O.rope_periods(256, 4, 100.0)
    for n in range(8):
        ks = 1 if n % 2 == 0 else 3
        B, h = int(rng.randint(1, 4)), int(rng.randint(1, 10))
        w = int(rng.randint(1, 8)) * (2 if ks == 3 else 1)
        H, W = 16 * h, 16 * w
        xd, st_in, gw, gb, wp, bias = _layer_inputs(dev, B, H, W, ks, 1000 + n)
        ty, tx = ops.rope_tables(per.to(dev), H, W)
        cat = torch.zeros((B, H, W, 256), dtype=torch.bfloat16, device=dev)
        keys = torch.zeros((B, h, w, 256), dtype=torch.bfloat16, device=dev)
        br = n % 2
        ops.stem_conv(xd, st_in, gw, gb, 1e-5, wp, bias, cat[..., 128 * br:128 * br + 128], None,
                      keys=(keys[..., 128 * br:128 * br + 128], ty, tx))
        y = cat[..., 128 * br:128 * br + 128].float().cpu().permute(0, 3, 1, 2).contiguous()
        ref = O.key_pool(O.rope(y, per, 2), (h, w))
        got = keys[..., 128 * br:128 * br + 128].float().cpu().permute(0, 3, 1, 2)
        err = (got - ref).abs()
        assert bool((err <= 1e-5 + 2 ** -8 * ref.abs()).all()), f"ks={ks} B={B} {H}x{W}: max err {float(err.max()):.3e}"


def test_forward_with_and_without_key_fusion_agree(dev):
    """naf_forward with the keys riding on the last stem layers (default) against the same forward with the separate pooling
    pre-pass (NAF_KEYS_FUSE=0, a process-wide A/B knob: second process): the keys differ by fp32 summation order only, i.e. by
    at most one bf16 rounding, and the outputs by what that moves (far inside the parity tolerance)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from oracle import naf_oracle as O\n"
        "from naf_amd import NAF\n"
        "p = O.make_params(seed=45)\n"
        "m = NAF(kernel_size=7).eval(); m.load_state_dict(p, strict=True); m = m.cuda()\n"
        "img = O.hash_normal((2, 3, 128, 160), 981).cuda(); ft = O.hash_normal((2, 128, 8, 10), 982).cuda().to(torch.bfloat16)\n"
        "with torch.no_grad(): o = m(img, ft, (128, 160))\n"
        "torch.save(o.float().cpu(), sys.argv[1])\n")
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for fuse in (None, "0"):
            env = dict(os.environ, NAF_HIP_KNOBS="1")
            env.pop("NAF_KEYS_FUSE", None)
            if fuse is not None:
                env["NAF_KEYS_FUSE"] = fuse
            f = os.path.join(td, f"o{fuse}.pt")
            r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(f))
    err = (outs[0] - outs[1]).abs()
    assert float(err.max()) <= 1.2e-2 and float(err.mean()) <= 2e-4, (float(err.max()), float(err.mean()))
