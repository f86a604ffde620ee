"""The differentiable HIP stem (naf_amd.model._HipStem: train.py:127-137 through convolutions.py:52-92) against torch autograd
on the same layers: every building block alone (plain convolution = data gradient, SiLU(GroupNorm) forward / backward with and
without the reflect-padding fold), then the gradients of all 36 encoder parameters and of the image through the whole model."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def group_stats(x):
    """fp64 {sum, sum^2} per (sample, group of C / 8 channels) of a [B, H, W, C] tensor, as the forward kernels accumulate them."""
    from naf_amd import ops
    B, C = x.shape[0], x.shape[-1]
    xd = x.double().reshape(B, -1, 8, C // 8)
    return ops.stats_from_total(torch.stack([xd.sum((1, 3)), (xd * xd).sum((1, 3))], dim=-1).contiguous())


@pytest.mark.parametrize("k,H,W,C", [(3, 24, 40, 128), (1, 16, 32, 128), (3, 7, 33, 128), (1, 5, 9, 128),
                                     (3, 11, 21, 48), (1, 9, 17, 96), (3, 10, 18, 256), (3, 2, 2, 16)])
def test_plain_convolution(dev, k, H, W, C):
    """naf_stem_conv_fwd without GroupNorm / SiLU == F.conv2d (reflect padding) on the bf16 inputs, fp32 accumulation; round 6: at
    every width the stem serves (the data gradient of the denoising models' layers), not only 128."""
    from naf_amd import ops
    g = torch.Generator(device="cpu").manual_seed(k * 100 + H)
    x = torch.randn(2, H, W, C, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(C, C, k, k, generator=g) / (C * k * k) ** 0.5).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    wp = ops.pack_conv_weight(w)
    y = torch.empty_like(x)
    ops.stem_conv_plain(x, wp, y, bias=b)
    xin = x.float().permute(0, 3, 1, 2)
    if k == 3:
        xin = F.pad(xin, (1, 1, 1, 1), mode="reflect")
    ref = F.conv2d(xin, ops.unpack_conv_weight(wp), b).permute(0, 2, 3, 1)
    err = (y.float() - ref).abs()
    assert float((err - 2 ** -8 * ref.abs()).max()) < 2e-3, float(err.max())     # one bf16 rounding of the output
    y2 = torch.empty_like(x)
    ops.stem_conv_plain(x, wp, y2)                                                # bias is optional
    assert rel(y2.float(), ref - b) < 5e-3


@pytest.mark.parametrize("pad,H,W", [(0, 12, 20), (1, 12, 20), (1, 2, 5)])
def test_act_forward(dev, pad, H, W):
    from naf_amd import ops
    g = torch.Generator(device="cpu").manual_seed(7 + pad)
    x = (torch.randn(2, H, W, 128, generator=g) * 2 + 0.5).to(dev).to(torch.bfloat16)
    gw, gb = (1 + 0.3 * torch.randn(128, generator=g)).to(dev), (0.2 * torch.randn(128, generator=g)).to(dev)
    a = ops.stem_act(x, group_stats(x), gw, gb, 1e-5, pad=pad)
    ref = F.silu(F.group_norm(x.float().permute(0, 3, 1, 2), 8, gw, gb, 1e-5))
    if pad:
        ref = F.pad(ref, (1, 1, 1, 1), mode="reflect")
    ref = ref.permute(0, 2, 3, 1)
    assert a.shape == ref.shape
    assert float((a.float() - ref).abs().max()) < 3e-2 and rel(a.float(), ref) < 4e-3


@pytest.mark.parametrize("fold,H,W,C", [(False, 12, 20, 128), (True, 12, 20, 128), (True, 3, 4, 128), (True, 2, 2, 128),
                                        (True, 9, 14, 48), (False, 7, 11, 96), (True, 6, 10, 240), (False, 5, 5, 16)])
def test_act_backward(dev, fold, H, W, C):
    """dx, d gamma, d beta of a = SiLU(GroupNorm(x)) vs autograd; with fold the incoming gradient lives on the reflect-padded
    domain and autograd differentiates through F.pad(mode='reflect') as well.  Widths whose 8-channel chunks straddle GroupNorm
    groups (48: groups of 6, 240: groups of 30, 16: groups of 2) are the denoising models' (round 6)."""
    from naf_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11 + H)
    B = 2
    x = (torch.randn(B, H, W, C, generator=g) * 1.5 - 0.3).to(dev).to(torch.bfloat16)
    gw, gb = (1 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    shp = (B, H + 2, W + 2, C) if fold else (B, H, W, C)
    da = torch.randn(shp, generator=g).to(dev).to(torch.bfloat16)
    dx = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=dev)
    sums = ops.stem_act_bwd(da, x, group_stats(x), gw, gb, 1e-5, dx, fold=fold)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    gwr, gbr = gw.clone().requires_grad_(True), gb.clone().requires_grad_(True)
    a = F.silu(F.group_norm(xr, 8, gwr, gbr, 1e-5))
    if fold:
        a = F.pad(a, (1, 1, 1, 1), mode="reflect")
    a.backward(da.float().permute(0, 3, 1, 2))
    assert rel(dx.float(), xr.grad.permute(0, 2, 3, 1)) < 6e-3
    assert rel(sums[..., 1].sum(0).float(), gwr.grad) < 2e-3 and rel(sums[..., 0].sum(0).float(), gbr.grad) < 2e-3


@pytest.mark.parametrize("k,B,H,W,C", [(3, 1, 20, 64, 128), (1, 2, 9, 32, 128), (3, 2, 13, 45, 128), (1, 1, 7, 19, 128), (3, 1, 2, 2, 128),
                                       (3, 2, 13, 45, 48), (1, 1, 9, 33, 96), (3, 1, 17, 40, 256), (3, 1, 6, 70, 144), (1, 2, 5, 5, 16),
                                       # the pipelined kernel (round 6): ranges cut inside image rows (commit-only extension segments at either end), last
                                       # segments of 22 / 4 / 13 pixels, odd and even segment counts per workgroup, one-segment workgroups, batch 2
                                       (3, 1, 200, 96, 128), (3, 2, 100, 150, 128), (3, 1, 40, 36, 128), (1, 1, 300, 160, 128), (3, 1, 64, 448, 128),
                                       (1, 2, 64, 100, 128), (3, 1, 173, 77, 128), (3, 1, 33, 35, 128), (3, 3, 31, 64, 128)])
def test_weight_gradient(dev, k, B, H, W, C):
    """naf_stem_wgrad vs autograd of conv(reflect_pad(SiLU(GroupNorm(x)))) w.r.t. the weight, on the same bf16 tensors; the widths
    other than 128 run stem_generic_bwd.hip (round 6: 144 and 256 split their output channels over two workgroups)."""
    from naf_amd import ops
    g = torch.Generator(device="cpu").manual_seed(31 + H)
    x = (torch.randn(B, H, W, C, generator=g) * 1.3 + 0.2).to(dev).to(torch.bfloat16)
    dy = torch.randn(B, H, W, C, generator=g).to(dev).to(torch.bfloat16)
    gw, gb = (1 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    dw, db = ops.stem_wgrad(dy, x, group_stats(x), gw, gb, 1e-5, k, with_bias=True)
    assert rel(db, dy.float().sum((0, 1, 2))) < 1e-4
    a = F.silu(F.group_norm(x.float().permute(0, 3, 1, 2), 8, gw, gb, 1e-5)).to(torch.bfloat16).float()   # the kernel's bf16 operand
    if k == 3:
        a = F.pad(a, (1, 1, 1, 1), mode="reflect")
    w = torch.zeros(C, C, k, k, device=dev, requires_grad=True)
    F.conv2d(a, w).backward(dy.float().permute(0, 3, 1, 2))
    assert dw.shape == w.grad.shape
    assert rel(dw, w.grad) < 3e-3, rel(dw, w.grad)
    a0 = ops.stem_act(x, group_stats(x), gw, gb, 1e-5, pad=0)                       # the two-call sequence: activation once, ...
    dw2 = ops.stem_wgrad(dy, a0, None, None, None, 1e-5, k)                          # ... plain pixel-contraction GEMM on it
    assert rel(dw2, w.grad) < 3e-3
    for t in range(k * k):       # every tap on its own: a wrong pixel shift of one tap must not hide in the norm of the others
        assert rel(dw[:, :, t // k, t % k], w.grad[:, :, t // k, t % k]) < 5e-3, t


@pytest.mark.parametrize("k,B,H,W,C", [(3, 2, 12, 20, 128), (1, 1, 9, 33, 128), (3, 1, 2, 2, 128), (3, 2, 11, 19, 48), (1, 1, 8, 8, 256), (3, 1, 5, 6, 16),
                                       # round 6, the matrix-pipe kernel: several segments per workgroup, a last segment of 4 / 13 pixels, every channel-tile count
                                       (3, 1, 70, 100, 128), (1, 2, 64, 96, 128), (3, 1, 40, 45, 160), (3, 2, 150, 77, 128), (1, 1, 90, 64, 96), (3, 1, 33, 200, 256),
                                       (3, 1, 24, 36, 32)])
def test_first_convolution_gradients(dev, k, B, H, W, C):
    """naf_stem_conv0_wgrad (weight, bias) and naf_stem_conv0_dgrad (image; round 6) vs autograd of Conv2d(3 -> C, reflect)."""
    from naf_amd import ops
    g = torch.Generator(device="cpu").manual_seed(41 + H)
    image = torch.randn(B, 3, H, W, generator=g).to(dev)
    dy = torch.randn(B, H, W, C, generator=g).to(dev).to(torch.bfloat16)
    dw, db = ops.stem_conv0_wgrad(dy, image, k)
    w = (torch.randn(C, 3, k, k, generator=g) / (3 * k * k) ** 0.5).to(dev).requires_grad_(True)
    bb = torch.zeros(C, device=dev, requires_grad=True)
    im = image.clone().requires_grad_(True)
    xin = F.pad(im, (1, 1, 1, 1), mode="reflect") if k == 3 else im
    F.conv2d(xin, w, bb).backward(dy.float().permute(0, 3, 1, 2))
    assert dw.shape == w.grad.shape and rel(dw, w.grad) < 1e-4 and rel(db, bb.grad) < 1e-4
    dwb, dbb = ops.stem_conv0_wgrad(dy, image.to(torch.bfloat16), k)            # bf16 image
    assert rel(dwb, w.grad) < 1e-2
    dimg = torch.full((B, 3, H, W), float("nan"), device=dev)
    ops.stem_conv0_dgrad(dy, w.detach().contiguous(), dimg)                       # written ...
    assert rel(dimg, im.grad) < 1e-5, rel(dimg, im.grad)
    ops.stem_conv0_dgrad(dy, w.detach().contiguous(), dimg, accumulate=True)      # ... or added to (the stem's second branch)
    assert rel(dimg, 2 * im.grad) < 1e-5


def test_strided_views(dev):
    """The kernels take the strided views the backward hands them: interior of a zero-bordered buffer, channel halves."""
    from naf_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    B, H, W = 1, 10, 18
    big = torch.randn(B, H + 4, W + 4, 256, generator=g).to(dev).to(torch.bfloat16)
    x = big[:, 2:H + 2, 2:W + 2, 128:]
    gw, gb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    a = ops.stem_act(x, group_stats(x), gw, gb, 1e-5, pad=0)
    a2 = ops.stem_act(x.contiguous(), group_stats(x), gw, gb, 1e-5, pad=0)
    assert torch.equal(a, a2)
    da = torch.randn(B, H + 2, W + 2, 128, generator=g).to(dev).to(torch.bfloat16)
    out = torch.zeros(B, H + 4, W + 4, 128, dtype=torch.bfloat16, device=dev)
    s1 = ops.stem_act_bwd(da, x, group_stats(x), gw, gb, 1e-5, out[:, 2:H + 2, 2:W + 2], fold=True)
    ref = torch.empty(B, H, W, 128, dtype=torch.bfloat16, device=dev)
    s2 = ops.stem_act_bwd(da.clone(), x.contiguous(), group_stats(x), gw, gb, 1e-5, ref, fold=True)
    assert torch.equal(out[:, 2:H + 2, 2:W + 2], ref) and float(out[:, :2].abs().max()) == 0.0
    assert rel(s1, s2) < 1e-6


@pytest.mark.parametrize("shape,lr", [((2, 256, 32, 48), (4, 6)), ((1, 256, 23, 30), (5, 7)), ((1, 128, 16, 16), (16, 16))])
def test_rope_pool_backward(dev, shape, lr):
    """naf_rope_pool_bwd vs autograd through the oracle's RoPE + adaptive key pooling (divisible, overlapping windows, ratio 1)."""
    from oracle import naf_oracle as O
    from naf_amd import ops
    B, C, H, W = shape
    heads = C // 64
    g = torch.Generator(device="cpu").manual_seed(21)
    per = O.rope_periods(C, heads, 100.0)
    x = torch.randn(shape, generator=g).requires_grad_(True)
    gq = torch.randn(shape, generator=g).to(torch.bfloat16).float()
    gk = torch.randn(B, C, *lr, generator=g)
    q = O.rope(x, per, heads)
    k = O.key_pool(q, lr)
    ((q * gq).sum() + (k * gk).sum()).backward()
    ty, tx = ops.rope_tables(per.to(dev), H, W)
    to5 = lambda t: t.to(dev).reshape(B, heads, 64, *t.shape[-2:]).permute(0, 1, 3, 4, 2).contiguous()
    dx = ops.rope_pool_bwd(to5(gq).to(torch.bfloat16), to5(gk), ty, tx, (H, W))
    assert dx.shape == shape and dx.dtype == torch.bfloat16
    assert rel(dx.float().cpu(), x.grad) < 4e-3


@pytest.mark.parametrize("size,lr,ksz,img_grad", [(96, 6, 3, True), (64, 4, 3, False)])
def test_model_gradients_match_torch_stem(dev, size, lr, ksz, img_grad):
    """forward_train(amp='hip') vs forward_train(amp=False) (fp32 torch stem, same HIP attention): output and the gradients of
    every encoder parameter (and of the image) agree to bf16-activation accuracy."""
    from naf_amd import NAF
    torch.manual_seed(3)
    model = NAF(kernel_size=ksz).to(dev).eval()          # eval: deterministic RoPE coordinates on both paths
    with torch.no_grad():                                  # non-trivial GroupNorm affine / biases
        for n, p in model.named_parameters():
            if n.endswith("bias") or "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    g = torch.Generator(device="cpu").manual_seed(9)
    image = torch.randn(2, 3, size, size, generator=g).to(dev)
    feats = torch.randn(2, 64, lr, lr, generator=g).to(dev)
    wout = torch.randn(2, 64, size, size, generator=g).to(dev)
    res = {}
    for mode in (False, "hip"):
        model.zero_grad(set_to_none=True)
        im = image.clone().requires_grad_(img_grad)
        out = model.forward_train(im, feats, (size, size), amp=mode)
        (out.float() * wout).sum().backward()
        res[mode] = (out.detach().float(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                     im.grad.detach().clone() if img_grad else None)
    out_r, gr, gi_r = res[False]
    out_h, gh, gi_h = res["hip"]
    assert rel(out_h, out_r) < 2e-2
    assert set(gr) == set(gh) and len(gr) == 36
    worst = max((rel(gh[n], gr[n]), n) for n in gr)
    assert worst[0] < 6e-2, worst
    if img_grad:
        assert rel(gi_h, gi_r) < 6e-2


def test_autocast_training_call_uses_the_hip_stem(dev, monkeypatch):
    """train.py:120-137 as the reference writes it: ``with torch.autocast(bfloat16): out = naf(image, feats, size)`` in
    .train() mode, loss.backward(), optimizer step -- dispatches to the differentiable HIP stem and trains."""
    from naf_amd import NAF, ops
    torch.manual_seed(0)
    model = NAF(kernel_size=3).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    calls = {"n": 0}
    real = ops.stem_conv_plain

    def spy(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    monkeypatch.setattr(ops, "stem_conv_plain", spy)
    image = torch.randn(1, 3, 64, 64, device=dev)
    feats = torch.randn(1, 32, 4, 4, device=dev)
    target = torch.randn(1, 32, 64, 64, device=dev)
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(image, feats, (64, 64))
        loss = (out.float() - target).pow(2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert calls["n"] == 3 * 8                       # eight 128 -> 128 layers, one data-gradient launch each
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    assert out.shape == (1, 32, 64, 64) and all(l == l for l in losses) and losses[-1] != losses[0]   # parameters moved


@pytest.mark.parametrize("img_hw,out_hw,lr_hw", [((64, 96), (64, 96), (4, 6)), ((128, 128), (64, 64), (4, 4))])
def test_hip_training_path_other_geometries(dev, img_hw, out_hw, lr_hw):
    """Non-square images, and an image larger than the output (the guidance is pooled between stem and RoPE, naf.py:34):
    forward_train(amp='hip') against the fp32 torch stem, outputs and encoder gradients."""
    from naf_amd import NAF
    torch.manual_seed(5)
    model = NAF(kernel_size=3).to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(13)
    image = torch.randn(1, 3, *img_hw, generator=g).to(dev)
    feats = torch.randn(1, 32, *lr_hw, generator=g).to(dev)
    wout = torch.randn(1, 32, *out_hw, generator=g).to(dev)
    res = {}
    for mode in (False, "hip"):
        model.zero_grad(set_to_none=True)
        out = model.forward_train(image, feats, out_hw, amp=mode)
        (out.float() * wout).sum().backward()
        res[mode] = (out.detach().float(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert rel(res["hip"][0], res[False][0]) < 2e-2
    worst = max((rel(res["hip"][1][n], res[False][1][n]), n) for n in res[False][1])
    assert worst[0] < 6e-2, worst


@pytest.mark.parametrize("dim,heads,ksz,C", [(256, 4, 3, 32), (96, 1, 5, 3), (512, 4, 3, 32), (160, 1, 3, 8)])
def test_training_call_without_autocast_runs_the_hip_stem(dev, dim, heads, ksz, C):
    """``model(image, feats, size)`` in .train() mode with NO autocast -- the reference's fp32 default (train.py:127-137,
    denoising.py:209-220: NAF(dim 96 ... 512, one head) on the noisy image itself) -- trains through the library's own differentiable
    stem since round 6, at every width the forward serves: a profiler table of the step holds no ATen / MIOpen
    convolution and no ATen GroupNorm, and the gradients agree with the explicit fp32 torch-stem arm (``amp=False``) to the
    bf16-activation budget the default width is held to."""
    from naf_amd import NAF
    torch.manual_seed(7)
    model = NAF(dim=dim, heads_attn=heads, heads_rope=heads, kernel_size=ksz).to(dev).train()
    model.image_encoder.rope.rescale_coords = None            # deterministic coordinates: the two arms see the same function
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias") or "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    S, lr = 48, 48 if C == 3 else 6
    g = torch.Generator(device="cpu").manual_seed(17)
    image = torch.randn(2, 3, S, S, generator=g).to(dev)
    feats = torch.randn(2, C, lr, lr, generator=g).to(dev)
    wout = torch.randn(2, C, S, S, generator=g).to(dev)
    res = {}
    for mode in ("call", False):
        model.zero_grad(set_to_none=True)
        im = image.clone().requires_grad_(True)
        if mode == "call":
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
                out = model(im, feats, (S, S))
                (out.float() * wout).sum().backward()
            names = {e.key for e in prof.key_averages()}
            bad = sorted(n for n in names if any(t in n.lower() for t in ("conv", "miopen", "group_norm")))
            assert not bad, bad
        else:
            out = model.forward_train(im, feats, (S, S), amp=False)
            (out.float() * wout).sum().backward()
        res[mode] = (out.detach().float(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                     im.grad.detach().clone())
    out_h, gh, gi_h = res["call"]
    out_r, gr, gi_r = res[False]
    assert rel(out_h, out_r) < 2e-2, rel(out_h, out_r)
    assert set(gr) == set(gh) and len(gr) == 36
    worst = max((rel(gh[n], gr[n]), n) for n in gr)
    assert worst[0] < 6e-2, worst
    assert rel(gi_h, gi_r) < 6e-2, rel(gi_h, gi_r)
