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
    st_in = torch.stack([g.sum(dim=(2, 3, 4)), (g * g).sum(dim=(2, 3, 4))], dim=-1).to(dev)
    wp = w.permute(2, 3, 0, 1).reshape(ks * ks, 128, 128).contiguous().to(torch.bfloat16).to(dev)
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
