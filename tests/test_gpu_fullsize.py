"""GPU parity at BASELINE.json's full sizes for the WHOLE forward (conv stem included), and through the torch.hub entry.

VERDICT r01 asked for: the whole forward checked at G1 size (the stem's strip / segment / XCD-remap geometry at 1024^2
used to run only inside bench.py), G3's 8-images-per-GPU shard through the sharded driver, a `-m gpu` test that goes
hubconf -> forward, and the offline half of SURVEY 8(f4): a reference-keyed checkpoint loaded through hubconf's own
`load_state_dict_from_url` path (a file:// URL works without a network) followed by a forward.

The oracle's conv stem at 1024^2 is ~1.4 TFLOP of fp32 CPU convolutions (tens of seconds on the GPU box's host); the
attention is compared on sampled rows (every column, so every 32-pixel strip border of the 3x3 kernel is covered; the
stem itself is compared on EVERY pixel).

Tolerances (fp, stated as the prompt asks): the bf16-activation stem against the fp32 oracle stem: mean |err| <= 8e-3,
max <= 2.5e-1 over 268 M values (profiles/r02_stem_error_budget.txt: every layer adds 2.9e-3 of its output RMS -- one bf16
rounding of the activated input and one of the output -- 6.9e-3 after five layers); whole forward: SURVEY.md section 8c's
|err| <= 2e-2 + 1e-2 |ref| elementwise (bf16 features and output where the workload says so) and mean |err| <= 6e-3.  The
attention averages the guidance error over its window, so the output error is an order of magnitude below the stem's.
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
    _lib.load()
    return torch.device("cuda:0")


def _load_model(dev, params, **kw):
    from naf_amd import NAF
    m = NAF(**kw).eval()
    m.load_state_dict(params, strict=True)
    return m.to(dev)


def _assert_close(got, ref, atol, rtol, what):
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {float(err.max()):.4e} at "
                                 f"{np.unravel_index(int(err.argmax()), err.shape)}, ref absmax {float(ref.abs().max()):.3f}")


def _oracle_rows(p, img, ft, out_sz, ksz, rows, heads=4, on=None):
    """Oracle forward of ONE image restricted to output rows ``rows`` (GroupNorm needs the whole image, so the stem and
    RoPE run at full size; only the attention is row-sampled).  Returns (stem [1,256,H,W], out_rows [1,C,len(rows),W]).
    ``on`` = a device: the same oracle code evaluated by ATen's fp32 device kernels (G4's 5.5 TFLOP stem and the G3 shard's two
    stems took a minute of host time; tests/test_gpu_parity.py::test_oracle_on_the_device_equals_the_oracle_on_the_host holds the
    device evaluation to the host's).  G1 -- the graded configuration -- and the G2 windows stay on the host."""
    if on is not None:
        p = {k: v.to(on) for k, v in p.items()}
        img, ft = img.to(on), ft.to(on)
    with torch.no_grad():
        stem = O.conv_stem(img, p)
        x = O.rope(stem, p["image_encoder.rope.periods"], heads)
        k = O.key_pool(x, ft.shape[-2:])
        iy = O.axis_index_table(out_sz, ft.shape[-2], ksz)[rows]
        ix = O.axis_index_table(out_sz, ft.shape[-1], ksz)
        ref = O.xna_tables(x[:, :, rows].contiguous(), k, ft, iy, ix, heads)
    return stem, ref                      # on `on` (or the host): the caller compares where the tensors live


def _assert_fused_keys(m, p, lr_hw, what, images=None, heads=4, on=None):
    """VERDICT r04 (weak 3): the keys the stem's LAST layers pooled (naf_stem_conv_keys_fwd inside the one-call forward), ALL cells at
    full size, against the oracle's pool(RoPE(.)) (naf.py:63-69 after rope.py:139-153) of the very bf16 guidance that call wrote
    -- both read back from the call's workspace (ForwardPlan.view) -- at tests/test_gpu_keys.py's tolerance (one bf16 rounding)."""
    plan = m.__dict__["_plan_cache"][1]
    assert plan is not None, f"{what}: not the one-call forward"
    torch.cuda.synchronize()
    guide = plan.view("guidance")
    keys = plan.view("keys")
    sel = list(range(guide.shape[0])) if images is None else images
    per = p["image_encoder.rope.periods"]
    worst = 0.0
    for b in sel:
        y = guide[b:b + 1].float()
        y = (y.cpu() if on is None else y.to(on)).permute(0, 3, 1, 2).contiguous()
        with torch.no_grad():
            ref = O.key_pool(O.rope(y, per if on is None else per.to(on), heads), lr_hw).cpu()     # `on`: see _oracle_rows
        del y
        got = keys[b:b + 1].float().cpu().permute(0, 3, 1, 2)
        err = (got - ref).abs()
        bad = err > 1e-5 + 2 ** -8 * ref.abs()
        assert not bool(bad.any()), (f"{what} image {b}: {int(bad.sum())}/{bad.numel()} fused keys out of tolerance, max err {float(err.max()):.3e} at "
                                     f"{np.unravel_index(int(err.argmax()), err.shape)}")
        worst = max(worst, float(err.max()))
    return worst


def test_whole_forward_G1_full_size(dev):
    """BASELINE configs[1] end to end: 1x3x1024^2 image, 768x64^2 features -> 1024^2, window 7, bf16 features."""
    out_sz, C, lr, ksz = 1024, 768, 64, 7
    p = O.make_params(seed=21)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((1, 3, out_sz, out_sz), 2101)
    ft = O.hash_normal((1, C, lr, lr), 2102).to(torch.bfloat16).float()
    # rows: image borders, a cell border (15 | 16), the XCD band / segment borders of the stem at 1024 rows (multiples
    # of 128 and of 64), mid-image, plus a few arbitrary ones
    # (round 5: + 383 | 384 and 639 | 640, so that every 128-row segment border of the 3x3 key-pooling launch lies inside a sampled window)
    rows = sorted({0, 1, 2, 15, 16, 63, 64, 127, 128, 255, 256, 300, 383, 384, 511, 512, 639, 640, 767, 768, 895, 896, 1007, 1008, 1021, 1022, 1023})
    stem_ref, ref = _oracle_rows(p, img, ft, out_sz, ksz, rows)
    got_stem = m.image_encoder.guidance(img.to(dev), (out_sz, out_sz)).float().cpu()
    err = (got_stem - stem_ref).abs()
    assert float(err.mean()) <= 8e-3 and float(err.max()) <= 2.5e-1, ("stem at 1024^2", float(err.mean()), float(err.max()))
    # per 32-pixel strip and per 64-row band: no strip / band stands out (a geometry bug would)
    band = err.mean(dim=(0, 1)).view(16, 64, 32, 32).mean(dim=(1, 3))
    assert float(band.max()) <= 2.0 * float(err.mean()) + 1e-3, "stem error is not uniform over strips / bands"
    del got_stem, err
    out = m(img.to(dev), ft.to(dev).to(torch.bfloat16), (out_sz, out_sz))
    assert out.shape == (1, C, out_sz, out_sz) and out.dtype == torch.bfloat16
    got = out[:, :, rows].float().cpu()
    _assert_close(got, ref, 2e-2, 1e-2, "G1 whole forward, sampled rows")
    assert float((got - ref).abs().mean()) <= 6e-3
    assert m.__dict__["_plan_cache"][1].planned_streams() == 2                     # the default at 1024^2: branches side by side
    _assert_fused_keys(m, p, (lr, lr), "G1")


def _whole_forward_case(dev, name, out_sz, C, lr, ksz, seed, rows, oracle_on=None):
    """Stem at EVERY pixel (mean / max and uniformity over 32-pixel strips and 64-row bands: a geometry bug would make one
    stand out) + the whole bf16 forward on sampled rows, against the fp32 oracle."""
    p = O.make_params(seed=seed)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((1, 3, out_sz, out_sz), 100 * seed + 1)
    ft = O.hash_normal((1, C, lr, lr), 100 * seed + 2).to(torch.bfloat16).float()
    stem_ref, ref = _oracle_rows(p, img, ft, out_sz, ksz, rows, on=oracle_on)
    got_stem = m.image_encoder.guidance(img.to(dev), (out_sz, out_sz)).float()
    err = (got_stem.to(stem_ref.device) - stem_ref).abs()        # 268 M - 1 G values: compared where the oracle's result lives
    del got_stem, stem_ref
    assert float(err.mean()) <= 8e-3 and float(err.max()) <= 2.5e-1, (f"{name} stem", float(err.mean()), float(err.max()))
    band = err.mean(dim=(0, 1)).view(out_sz // 64, 64, out_sz // 32, 32).mean(dim=(1, 3))
    assert float(band.max()) <= 2.0 * float(err.mean()) + 1e-3, f"{name}: stem error is not uniform over strips / bands"
    del err, band
    out = m(img.to(dev), ft.to(dev).to(torch.bfloat16), (out_sz, out_sz))
    assert out.shape == (1, C, out_sz, out_sz) and out.dtype == torch.bfloat16
    got = out[:, :, rows].float().cpu()
    ref = ref.cpu()
    _assert_close(got, ref, 2e-2, 1e-2, f"{name} whole forward, sampled rows")
    assert float((got - ref).abs().mean()) <= 6e-3
    _assert_fused_keys(m, p, (lr, lr), name, on=oracle_on)        # every cell: the POOL launch's segment rounds at this size (4 x 128 rows at 2048^2)


def test_whole_forward_G4_full_size(dev):
    """VERDICT r02 (missing 3): BASELINE configs[4] end to end -- 1x3x2048^2 image, 768x128^2 features -> 2048^2, window 7,
    bf16 -- the stem's strip / segment geometry at 2048 rows (64 strips x 4 segments of 512 rows on 256 CUs) had never been
    compared with anything; the attention's bf16 output is 3.2 G elements (offsets past 2^31).  The oracle stem at 2048^2 is
    5.5 TFLOP of fp32 CPU convolutions: about a minute on the GPU box's host cores."""
    rows = sorted({0, 1, 15, 16, 511, 512, 1023, 1024, 1300, 1535, 1536, 2031, 2032, 2046, 2047})
    _whole_forward_case(dev, "G4", 2048, 768, 128, 7, 24, rows, oracle_on=dev)      # the oracle's code on the device (see _oracle_rows)


@pytest.mark.parametrize("ksz", [7, 11, 15])
def test_whole_forward_G2_full_size(dev, ksz):
    """BASELINE configs[2] end to end: 1x3x512^2 image, 1024x32^2 features -> 512^2, windows 7 / 11 / 15, bf16: the staged
    4-wave cell kernel (k = 7) and the bf16 sliding-window kernel (k = 11, 15) behind the real stem, rotate-on-load."""
    rows = sorted({0, 1, 15, 16, 127, 128, 255, 256, 400, 495, 496, 510, 511})
    _whole_forward_case(dev, f"G2-k{ksz}", 512, 1024, 32, ksz, 25 + ksz, rows)


def test_whole_forward_G3_shard_through_the_sharded_driver(dev):
    """BASELINE configs[3] as one rank sees it: 8 of the 64 images (C = 1024), run through naf_amd.dist.ShardedNAF in
    micro-batches exactly as bench.py --gpus 8 does; images 0 and 7 of the shard against the oracle on sampled rows."""
    from naf_amd import dist as nd
    out_sz, C, lr, ksz, nb = 1024, 1024, 64, 7, 8
    p = O.make_params(seed=22)
    m = _load_model(dev, p, kernel_size=ksz)
    img = O.hash_normal((nb, 3, out_sz, out_sz), 2201)
    ft = O.hash_normal((nb, C, lr, lr), 2202).to(torch.bfloat16).float()
    out = nd.ShardedNAF(m, micro_batch=2)(img.to(dev), ft.to(dev).to(torch.bfloat16), (out_sz, out_sz))
    assert out.shape == (nb, C, out_sz, out_sz) and out.dtype == torch.bfloat16
    rows = sorted({0, 16, 511, 512, 1023})
    for b in (0, 7):
        _, ref = _oracle_rows(p, img[b:b + 1], ft[b:b + 1], out_sz, ksz, rows, on=dev)   # the oracle's code on the device (G1 keeps the host)
        ref = ref.cpu()
        got = out[b:b + 1, :, rows].float().cpu()
        _assert_close(got, ref, 2e-2, 1e-2, f"G3 shard image {b}, sampled rows")
        assert float((got - ref).abs().mean()) <= 6e-3
    # the last micro-batch (images 6, 7: two images per launch, eight rounds of 128-row segments in the 3x3 key-pooling launch)
    # is still in the plan's workspace: all of its cells against the oracle
    _assert_fused_keys(m, p, (lr, lr), "G3 shard, last micro-batch", on=dev)
    # micro-batching is invisible up to the order of the GroupNorm partial sums (the stem's per-workgroup fp32 partials are
    # cut differently for 1 and 2 images per launch, the fp64 atomics land in any order): a handful of bf16 roundings flip
    alone = m(img[3:4].to(dev), ft[3:4].to(dev).to(torch.bfloat16), (out_sz, out_sz))
    d = (alone[0].float() - out[3].float()).abs()
    assert float(d.max()) <= 3.2e-2 and float(d.mean()) <= 2e-4, (float(d.max()), float(d.mean()))


def _state_for_oracle(model):
    return {k: v.detach().float().cpu() for k, v in model.state_dict().items()}


def test_hub_entry_forward_on_gpu(dev):
    """hubconf.naf(pretrained=False, device="cuda") -> naf(image, lr_features, target_size) (hubconf.py:8-24, README
    usage) against the oracle run on the very weights the hub call initialised."""
    import hubconf
    torch.manual_seed(1234)
    naf = hubconf.naf(pretrained=False, device="cuda")
    naf.eval()                                           # README usage (README.md:105-121)
    assert next(naf.parameters()).is_cuda and naf.upsampler.kernel_size == (9, 9)
    p = _state_for_oracle(naf)
    img = O.hash_normal((1, 3, 144, 176), 2301)
    ft = O.hash_normal((1, 384, 9, 11), 2302)
    out = naf(img.to(dev), ft.to(dev), (144, 176))
    ref = O.naf_forward(p, img, ft, (144, 176), kernel_size=9)
    got = out.float().cpu()
    _assert_close(got, ref, 2e-2, 1e-2, "hub entry forward")
    assert float((got - ref).abs().mean()) <= 6e-3


def test_hub_pretrained_checkpoint_from_a_local_url(dev, tmp_path, monkeypatch):
    """SURVEY 8(f4), offline half: a checkpoint with the reference's state_dict keys (naf_release.pth layout, strict)
    is fetched through hubconf's own torch.hub.load_state_dict_from_url call -- pointed at a file:// URL -- and the
    loaded model reproduces the oracle with those weights."""
    import hubconf
    p = O.make_params(seed=33)                       # reference-keyed: image_encoder.{encoder,sem_encoder}.*, rope.periods
    ckpt = tmp_path / "naf_release.pth"
    torch.save({k: v.clone() for k, v in p.items()}, ckpt)
    monkeypatch.setattr(hubconf, "CHECKPOINT_URL", ckpt.as_uri())
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "hub"))
    naf = hubconf.naf(pretrained=True, device="cuda").eval()
    got_sd = naf.state_dict()
    assert set(got_sd) == set(p) and all(torch.equal(got_sd[k].cpu(), p[k]) for k in p)
    img = O.hash_normal((2, 3, 96, 96), 2401)
    ft = O.hash_normal((2, 192, 12, 12), 2402)           # 9 x 9 window at dilation 8: 72 <= 96 (NATTEN's precondition)
    out = naf(img.to(dev), ft.to(dev), [96, 96])
    ref = O.naf_forward(p, img, ft, (96, 96), kernel_size=9)
    _assert_close(out.float().cpu(), ref, 2e-2, 1e-2, "pretrained-from-file forward")
    # a checkpoint with a missing key must fail loudly (strict load, as the reference's load_state_dict does)
    bad = {k: v for k, v in p.items() if not k.endswith("rope.periods")}
    torch.save(bad, tmp_path / "bad.pth")
    monkeypatch.setattr(hubconf, "CHECKPOINT_URL", (tmp_path / "bad.pth").as_uri())
    with pytest.raises(RuntimeError):
        hubconf.naf(pretrained=True, device="cuda")


def test_feature_provider_hook_feeds_vit_shaped_tokens(dev):
    """the backbone side of the call (src/backbone/vit_wrapper.py:46-180 returns NCHW patch features from
    a ViT's token sequence).  A synthetic ViT-shaped source -- [B, 1 + n_reg + h*w, C] tokens with a class token and
    register tokens in front, patch 14 -- goes through naf_amd.features.tokens_to_feature_map and into the forward."""
    from naf_amd import features
    B, C, hp, wp, nreg = 1, 384, 16, 16, 4
    torch.manual_seed(7)
    tokens = torch.randn(B, 1 + nreg + hp * wp, C, device=dev)
    fmap = features.tokens_to_feature_map(tokens, (hp, wp), num_prefix_tokens=1 + nreg)
    assert fmap.shape == (B, C, hp, wp)
    assert torch.equal(fmap[0, :, 3, 5], tokens[0, 1 + nreg + 3 * wp + 5])
    provider = features.SyntheticViT(embed_dim=C, patch_size=14, num_prefix_tokens=1 + nreg, seed=3).to(dev)
    img = torch.randn(B, 3, hp * 14, wp * 14, device=dev)
    lr = provider(img)
    assert lr.shape == (B, C, hp, wp)
    p = O.make_params(seed=34)
    m = _load_model(dev, p, kernel_size=9)
    out = m(img, lr, (hp * 14, wp * 14))
    ref = O.naf_forward(p, img.cpu(), lr.float().cpu(), (hp * 14, wp * 14), kernel_size=9)
    _assert_close(out.float().cpu(), ref, 2e-2, 1e-2, "ViT-token features through the forward")


def test_captured_forward_survives_other_shapes_and_cache_turnover(dev):
    """a hipGraph replay dereferences the forward plan's workspace, the RoPE tables and the packed
    weights, which live in single-slot caches of the model.  Capture, run eager forwards with OTHER shapes (which
    replace every one of those slots), drop the allocator's cached blocks, replay: the result must still be exact."""
    p = O.make_params(seed=41)
    m = _load_model(dev, p, kernel_size=3)
    img = O.hash_normal((1, 3, 64, 64), 4101).to(dev)
    ft = O.hash_normal((1, 128, 4, 4), 4102).to(dev).to(torch.bfloat16)
    want = m(img, ft, (64, 64)).clone()
    g = m.capture(img, ft, (64, 64))
    assert torch.equal(g(), want)
    for hw, lr in (((96, 80), (6, 5)), ((48, 48), (3, 3)), ((128, 128), (8, 8))):          # turn every cache slot over
        im2 = O.hash_normal((2, 3, *hw), 4103).to(dev)
        f2 = O.hash_normal((2, 64, *lr), 4104).to(dev)
        m(im2, f2, hw)
        del im2, f2
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(8)]              # recycle whatever was freed
    torch.cuda.synchronize()
    assert torch.equal(g(), want)
    img2 = O.hash_normal((1, 3, 64, 64), 4105).to(dev)
    want2 = m(img2, ft, (64, 64)).clone()
    assert torch.equal(g(img2), want2)
    del junk


def test_forward_is_differentiable_when_a_gradient_is_wanted(dev):
    """the reference's forward is differentiable (train.py:127-137, denoising.py:213 call model(...) and
    backprop).  naf(...) takes the fused inference path under no_grad / in eval mode and dispatches to forward_train when
    autograd is on and an input requires grad or the module is in train mode."""
    p = O.make_params(seed=42)
    m = _load_model(dev, p, kernel_size=3)
    img = O.hash_normal((1, 3, 32, 32), 4201).to(dev)
    ft = O.hash_normal((1, 64, 4, 4), 4202).to(dev)
    out = m(img, ft, (32, 32))                                   # eval mode: inference kernels, no graph
    assert out.grad_fn is None
    m.train()
    m.image_encoder.rope.rescale_coords = None                   # keep the comparison deterministic
    out_t = m(img, ft, (32, 32))
    assert out_t.grad_fn is not None
    out_t.float().square().mean().backward()
    g = m.image_encoder.sem_encoder[0].weight.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0
    with torch.no_grad():
        assert m(img, ft, (32, 32)).grad_fn is None              # train mode under no_grad: inference path
    m.eval()
    ftg = ft.clone().requires_grad_(True)
    assert m(img, ftg, (32, 32)).grad_fn is not None             # eval mode, but a gradient w.r.t. the features is wanted
    # both paths compute the same function (fp32 torch stem vs bf16 fused stem: loose tolerance)
    _assert_close(out_t.detach().float().cpu(), out.float().cpu(), 2e-2, 1e-2, "forward_train vs inference path")   # measured 7.3e-3


@pytest.mark.parametrize("launcher,world,total", [("torch.distributed.run", 2, 6), ("plain", 8, 12)])
def test_bench_multi_rank_path_dry_run(dev, launcher, world, total):
    """bench.py --gpus N as the driver launches it (torch.distributed.run, one rank per process; 2 ranks) AND as the plain command
    line `python bench.py --gpus 8` (bench.py then starts its ranks itself on a free port) -- the world size the driver's scaling run
    ends at, with 12 images so that the shards are UNEVEN (2,2,2,2,1,1,1,1: scatter padding, a micro-batch larger than a shard) -- on
    ONE GPU with the collectives on gloo (NAF_BENCH_BACKEND=gloo: a rehearsal of the code path, never a measurement): the G3
    experiment -- rank 0 owns the batch, parameters by flat broadcast, inputs by scatter, every rank runs its shard through
    ShardedNAF, rank 0 then runs the whole batch alone for speedup_vs_1 -- prints one well-formed JSON line."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NAF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tail = [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--total-batch", str(total)]
    if launcher == "plain":
        cmd = [sys.executable, *tail]
        env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), *tail]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    shard0 = -(-total // world)                                  # rank 0 holds the largest shard (block partition)
    assert j["n_gpus"] == world and j["scaling"] == "strong" and j["config"]["global_batch"] == total and j["config"]["per_gpu_batch"] == shard0
    assert j["config"]["workload"].startswith("G3") and "DRY RUN" in j["data"] and "speedup_vs_1" in j["config"]["scale_note"]
    assert j["value"] > 0 and j["one_gpu_ms"] > 0 and j["speedup_vs_1"] > 0 and j["scatter_ms"] > 0
    assert j["roofline"]["kernel_ms"] > 0 and j["cpu_baseline"] is None
    # per-step device times of both readings (round 6): as many samples as steps, the first one and the spread on record
    for key in ("step_ms", "step_ms_no_settle"):
        assert j[key]["samples"] == 2 and j[key]["first"] > 0 and j[key]["min"] <= j[key]["median"] <= j[key]["max"]
    # the line proves what ran where: one entry per rank with the device it computed on and its own step time (under RCCL
    # the identities must be distinct -- bench.py exits otherwise; in this dry run the ranks share the GPU), a weak-scaling
    # leg beside the strong one, and the phases of the single call
    assert [r["rank"] for r in j["ranks"]] == list(range(world)) and all(r["ms_per_step"] > 0 and r["name"] for r in j["ranks"])
    assert sum(r["images"] for r in j["ranks"]) == total and max(r["images"] for r in j["ranks"]) - min(r["images"] for r in j["ranks"]) <= 1
    assert all("uuid" in r and "pci" in r for r in j["ranks"])
    assert j["weak_leg"]["images_per_gpu"] == min(8, total // world) and j["weak_leg"]["ms_all_ranks_busy"] > 0 and j["weak_leg"]["ms_rank0_alone"] > 0
    ph = j["phases_ms"]
    assert ph["stem"] > 0 and ph["rope_pool"] > 0 and ph["attention"] > 0 and ph["stem_conv3"] > 0
    # (<=: with eight processes time-slicing one GPU the attention launch takes so long that the 5 us pre-pass vanishes in the fourth decimal)
    assert 0 < j["roofline"]["frac_with_prepass"] <= j["roofline"]["frac"]


@pytest.mark.parametrize("amp", ["auto", False])
def test_denoising_configuration_trains_on_the_matrix_core_backward(dev, amp):
    """denoising.py:209-220's step -- model(noisy_norm, noisy, (S, S)) in train mode, loss, backward -- at NAF(dim 96, one head,
    window 15): the attention's backward runs xna_rows_bwd_kernel (round 3; the scalar table-driven kernel before), and every
    parameter gradient plus the gradient w.r.t. the noisy image agrees with fp32 autograd through the oracle."""
    from naf_amd import ops
    dim, S = 96, 40
    p = O.make_params(dim=dim, heads_rope=1, seed=72)
    m = _load_model(dev, p, dim=dim, heads_attn=1, heads_rope=1, kernel_size=15)
    img = O.hash_normal((1, 3, S, S), 7201)
    noisy = O.hash_normal((1, 3, S, S), 7202)
    wgt = O.hash_normal((1, 3, S, S), 7203)
    po = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "periods" not in k) for k, v in p.items()}
    no = noisy.clone().requires_grad_(True)
    (O.naf_forward(po, img, no, (S, S), kernel_size=15, heads_attn=1, heads_rope=1) * wgt).sum().backward()
    seen = []
    real = ops.xna_backward
    def spy(q, k, v, g, ks, **kw):
        seen.append(ops.xna_backward_select(q, k, v, ks))
        return real(q, k, v, g, ks, **kw)
    for prm in m.parameters():
        prm.requires_grad_(True)
    nd = noisy.to(dev).requires_grad_(True)
    ops.xna_backward = spy
    try:
        out = m.forward_train(img.to(dev), nd, (S, S), amp=amp)      # "auto" = what model(...) runs: the library's own stem at width 48 (round 6)
        (out.float() * wgt.to(dev)).sum().backward()
    finally:
        ops.xna_backward = real
    assert seen == ["rows"], seen
    # fp32 torch stem (amp=False): the bounds of rounds 3-5; the HIP stem ("auto"): bf16 activations between ten layers, the budget
    # tests/test_gpu_train_stem.py holds the default width to (6e-2 relative)
    tol_p, tol_f = (5e-2, 3e-2) if amp is False else (6e-2, 3e-2)      # measured 3.9e-2 / 1.3e-2 (gpurun call 634)
    checked, worst = 0, 0.0
    for name, prm in m.named_parameters():
        ref = po[name].grad
        if ref is None:
            continue
        got = prm.grad.float().cpu()
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err <= tol_p * scale + 1e-3, f"{name}: grad err {err:.3e} vs max {scale:.3e}"
        worst = max(worst, err / max(scale, 1e-30))
        checked += 1
    assert checked >= 20
    gs = float(no.grad.abs().max())
    ef = float((nd.grad.float().cpu() - no.grad).abs().max())
    print("denoising training step, amp=%s: worst parameter gradient %.3e, feature gradient %.3e (relative to the largest entry)" % (amp, worst, ef / gs))
    assert ef <= tol_f * gs + 1e-3


@pytest.mark.parametrize("dim,shape", [(96, (1, 40, 56)), (128, (2, 33, 47)), (512, (1, 24, 40)), (32, (1, 20, 24))])
def test_stem_of_any_width_runs_on_hip_and_matches_the_oracle(dev, dim, shape):
    """VERDICT r01 (missing 2): hidden widths other than 128 -- the reference's denoising models, NAF(dim = 96 ... 512)
    (denoising.py:213) -- used to drop to torch / MIOpen; they now run the general HIP kernels (stem_generic.hip).  Whole
    stem (both branches, five layers each) against the fp32 oracle stem, and layer by layer against torch ops fed with
    the HIP kernel's own bf16 input (isolates one layer's error)."""
    import torch.nn.functional as F
    from naf_amd import ops
    B, H, W = shape
    heads = 1 if dim % 256 else 4
    p = O.make_params(dim=dim, heads_rope=heads, seed=60 + dim)
    m = _load_model(dev, p, dim=dim, heads_attn=heads, heads_rope=heads, kernel_size=3)
    assert m.image_encoder._hip_stem_ok() and m.image_encoder.stem_impl == "hip"
    img = O.hash_normal((B, 3, H, W), 6000 + dim)
    ref = O.conv_stem(img, p)
    got = m.image_encoder._stem_hip(img.to(dev)).float().cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    assert float(err.mean()) <= 8e-3 and float(err.max()) <= 2.0e-1, (dim, float(err.mean()), float(err.max()))
    # one GroupNorm -> SiLU -> conv layer in isolation (3x3 and 1x1), same bf16 input on both sides
    hid = dim // 2
    x = O.hash_normal((B, hid, H, W), 6100 + dim).to(torch.bfloat16)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous()                                   # [B, H, W, hid]
    for ks, pre in ((3, "image_encoder.sem_encoder.1"), (1, "image_encoder.encoder.1")):
        w, bias = p[f"{pre}.conv1.weight"], p[f"{pre}.conv1.bias"]
        gw, gb = p[f"{pre}.norm1.weight"], p[f"{pre}.norm1.bias"]
        xf = x.float()
        st = ops.stats_from_total(torch.stack([xf.double().view(B, 8, -1).sum(-1), (xf.double() ** 2).view(B, 8, -1).sum(-1)], dim=-1).to(dev))
        y = torch.empty((B, H, W, hid), dtype=torch.bfloat16, device=dev)
        st_out = ops.new_stats(B, dev)
        wp = ops.pack_conv_weight(w).to(dev)
        ops.stem_conv(xd, st, gw.to(dev), gb.to(dev), 1e-5, wp, bias.to(dev), y, st_out)
        a = F.silu(F.group_norm(xf, 8, gw, gb, eps=1e-5)).to(torch.bfloat16).float()
        if ks == 3:
            a = F.pad(a, (1, 1, 1, 1), mode="reflect")
        r = F.conv2d(a, w.to(torch.bfloat16).float(), bias)
        g1 = y.permute(0, 3, 1, 2).float().cpu()
        _assert_close(g1, r, 3e-2, 1.6e-2, f"dim {dim} layer k{ks}")
        rs = torch.stack([r.double().view(B, 8, -1).sum(-1), (r.double() ** 2).view(B, 8, -1).sum(-1)], dim=-1)
        assert torch.allclose(ops.stats_total(st_out).cpu(), rs, rtol=2e-3, atol=0.5)


def test_denoising_configuration_runs_entirely_on_hip(dev):
    """The reference's denoising call (denoising.py:213,301: model(noisy_norm, noisy, (S, S)) with NAF(dim, heads 1,
    window 15): ratio 1, C = 3) at dim 96 -- stem on stem_generic.hip, attention on the matrix-core row kernel."""
    dim, S = 96, 48
    p = O.make_params(dim=dim, heads_rope=1, seed=71)
    m = _load_model(dev, p, dim=dim, heads_attn=1, heads_rope=1, kernel_size=15)
    img = O.hash_normal((1, 3, S, S), 7101)
    noisy = O.hash_normal((1, 3, S, S), 7102)
    out = m(img.to(dev), noisy.to(dev), (S, S)).float().cpu()
    ref = O.naf_forward(p, img, noisy, (S, S), kernel_size=15, heads_attn=1, heads_rope=1)
    # one head, a 15x15 window at ratio 1 (every query's own cell dominates its softmax), three value channels: measured maximum
    # 5.8e-2, 16 of 6912 elements outside SURVEY 8c's 2e-2 + 1e-2 |ref|, mean 1.4e-3 (profiles/r04_tolerance_budget.txt); asserted
    # with 1.5x head-room on the maximum and three times the outliers
    _assert_close(out, ref, 8.7e-2, 1e-2, "denoising configuration")
    assert float(((out - ref).abs() > 2e-2 + 1e-2 * ref.abs()).float().mean()) <= 7e-3
    assert float((out - ref).abs().mean()) <= 3e-3


def test_attention_G3_whole_batch_of_64_in_one_launch(dev):
    """BASELINE configs[3]'s whole batch on ONE GPU in ONE launch (maximum size: 64 images, C = 1024, 1024^2: 34 GB of queries,
    137 GB of bf16 output, 2^36 output elements -- every offset past 2^31 elements is exercised): partition of unity on
    every image, sampled rows of the first, a middle and the last image against the oracle."""
    from naf_amd import ops
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 200 * 2 ** 30:
        pytest.skip(f"needs ~175 GB of free device memory, {free / 2 ** 30:.0f} GB available")
    torch.cuda.empty_cache()
    B, C, lr, out_sz, ksz, heads = 64, 1024, 64, 1024, 7, 4
    gen = torch.Generator(device="cpu").manual_seed(4321)
    k = torch.randn(B, 256, lr, lr, generator=gen)
    v = torch.randn(B, C, lr, lr, generator=gen).to(torch.bfloat16).float()
    q5 = torch.empty((B, heads, out_sz, out_sz, 64), dtype=torch.bfloat16, device=dev)
    g = torch.Generator(device=dev).manual_seed(6)
    for b in range(B):                                    # per image: no 68 GB fp32 temporary
        q5[b] = torch.randn(heads, out_sz, out_sz, 64, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    k5 = k.view(B, heads, 64, lr, lr).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16).to(dev)
    vp = ops.pack_values(v.to(dev))
    v5 = vp.view(B, lr, lr, heads, C // heads).permute(0, 3, 1, 2, 4)
    out = ops.xna_forward(q5, k5, v5, ksz, out_dtype=torch.bfloat16, path="mfma")
    torch.cuda.synchronize()
    assert out.shape == (B, heads, out_sz, out_sz, C // heads)
    rows = [0, 17, 511, 1023]
    iy = O.axis_index_table(out_sz, lr, ksz)[rows]
    ix = O.axis_index_table(out_sz, lr, ksz)
    kb = k.to(torch.bfloat16).float()
    for b in (0, 31, 63):
        q_rows = q5[b:b + 1, :, rows].float().cpu().permute(0, 1, 4, 2, 3).reshape(1, 256, len(rows), out_sz)
        ref = O.xna_tables(q_rows, kb[b:b + 1], v[b:b + 1], iy, ix, heads)
        got = out[b:b + 1, :, rows].permute(0, 1, 4, 2, 3).reshape(1, C, len(rows), out_sz).float().cpu()
        _assert_close(got, ref, 1.2e-2, 1.2e-2, f"B=64 launch, image {b}")
    del out
    ones = ops.xna_forward(q5, k5, torch.full_like(v5, 0.5), ksz, out_dtype=torch.bfloat16, path="mfma")
    for b in range(0, B, 7):
        assert float((ones[b].float() - 0.5).abs().max()) <= 4e-3


@pytest.mark.parametrize("name,C,lr,out,ksz", [
    ("G1", 768, 64, 1024, 7),          # xna_bwd2_kernel<7, 192>: 256 runs of 64 cells, one per CU
    ("G2-k7", 1024, 32, 512, 7),       # xna_bwd2_kernel<7, 256>: one window buffer, a V key tile from the LDS per round
    ("G2-k11", 1024, 32, 512, 11),     # the eight-wave kernel in two channel chunks of 128
    ("G2-k15", 1024, 32, 512, 15),     # four channel chunks of 64 on the eight-wave kernel (dQ accumulated across launches)
])
def test_attention_backward_at_benched_sizes_against_the_scalar_kernel(dev, name, C, lr, out, ksz):
    """The attention backward at BASELINE's sizes (the oracle's autograd would take minutes there): the matrix-core cell kernels against the
    independent table-driven scalar kernel (fp32 throughout, one wave per query) on the same bf16 inputs, every element of dq / dk / dv."""
    from naf_amd import ops
    heads = 4
    gen = torch.Generator(device=dev).manual_seed(8800 + C + ksz)
    q = torch.randn(1, heads, out, out, 64, device=dev, generator=gen).to(torch.bfloat16)
    k = torch.randn(1, heads, lr, lr, 64, device=dev, generator=gen).to(torch.bfloat16)
    v = torch.randn(1, lr, lr, heads, C // heads, device=dev, generator=gen).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    g = torch.randn(1, out, out, heads, C // heads, device=dev, generator=gen).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    assert ops.xna_backward_select(q, k, v, ksz) == "mfma"
    a = ops.xna_backward(q, k, v, g, ksz)
    b = ops.xna_backward(q, k, v, g, ksz, path="generic")
    for x, y, nm in zip(a, b, ("dq", "dk", "dv")):
        scale = float(y.float().abs().max())
        err = (x.float() - y.float()).abs()
        assert bool(torch.isfinite(x.float()).all()), (name, nm)
        # dk / dv add up 256 queries x k^2 windows of bf16-rounded P / dS per key: the cell-fuzz tolerance
        assert float(err.max()) <= 2.5e-2 * scale + 1e-3 and float(err.mean()) <= 3e-3 * scale + 1e-4, (name, nm, float(err.max()), float(err.mean()), scale)


def test_edge_inputs_empty_batch_dtypes_and_strides(dev):
    """Edge cases at the operator boundary: an empty batch (empty tensors in, empty tensors out, as through the reference's torch
    ops), fp16 / fp64 features (computed through the bf16 / fp32 contract, returned in the caller's dtype), a non-contiguous
    image view and channels-last features, batch elements that are views of a larger tensor."""
    p = O.make_params(seed=81)
    m = _load_model(dev, p, kernel_size=3)
    out = m(torch.empty(0, 3, 32, 32, device=dev), torch.empty(0, 64, 2, 2, device=dev), (32, 32))
    assert out.shape == (0, 64, 32, 32)
    o2, lg = m(torch.empty(0, 3, 32, 32, device=dev), torch.empty(0, 64, 2, 2, device=dev, dtype=torch.bfloat16), (32, 32), return_weights=True)
    assert o2.shape == (0, 64, 32, 32) and o2.dtype == torch.bfloat16 and lg.shape == (0, 4, 32, 32, 9)
    img = O.hash_normal((2, 3, 48, 64), 8101)
    ft = O.hash_normal((2, 64, 3, 4), 8102)
    ref = O.naf_forward(p, img, ft, (48, 64), kernel_size=3)
    base = m(img.to(dev), ft.to(dev), (48, 64)).float().cpu()
    _assert_close(base, ref, 2e-2, 1e-2, "contiguous fp32 inputs")
    for dt in (torch.float16, torch.float64):
        o = m(img.to(dev), ft.to(dev).to(dt), (48, 64))
        assert o.dtype == dt
        _assert_close(o.float().cpu(), ref, 2e-2, 1e-2, f"{dt} features")
    # strided views: the image as a crop of a larger canvas, the features channels-last, both as slices of a bigger batch
    canvas = torch.zeros(4, 3, 60, 80, device=dev)
    canvas[1:3, :, 5:53, 7:71] = img.to(dev)
    fcl = torch.zeros(4, 3, 4, 64, device=dev)
    fcl[1:3] = ft.to(dev).permute(0, 2, 3, 1)
    o = m(canvas[1:3, :, 5:53, 7:71], fcl[1:3].permute(0, 3, 1, 2), (48, 64)).float().cpu()
    assert float((o - base).abs().max()) <= 1e-6, "strided inputs must give the contiguous result"
    with pytest.raises(ValueError):
        m(img.to(dev), ft[:1].to(dev), (48, 64))                       # batch mismatch
    with pytest.raises(ValueError):
        m(img.to(dev), ft.to(dev), (2, 64))                            # output smaller than the feature grid (dilation 0)


def _forward_hashes(cases, env_extra):
    """SHA-256 of the bf16 output of ``naf(image, feats, size)`` for every (Ho, Wo, lh, lw, C, k) of ``cases``, in a fresh process with
    ``env_extra`` set (the library reads its A/B knobs once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, torch; sys.path.insert(0, %r)\n"
        "from oracle import naf_oracle as O\n"
        "from naf_amd import NAF\n"
        "for (Ho, Wo, lh, lw, C, k) in %r:\n"
        "    p = O.make_params(seed=61)\n"
        "    m = NAF(kernel_size=k).eval(); m.load_state_dict(p, strict=True); m = m.cuda()\n"
        "    img = O.hash_normal((1, 3, Ho, Wo), 6101).cuda(); ft = O.hash_normal((1, C, lh, lw), 6102).cuda().to(torch.bfloat16)\n"
        "    out = m(img, ft, (Ho, Wo)); torch.cuda.synchronize()\n"
        "    print('SHA', Ho, Wo, k, hashlib.sha256(out.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest())\n" % (root, tuple(cases)))
    env = dict(os.environ, NAF_HIP_KNOBS="1", **env_extra)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [l for l in r.stdout.splitlines() if l.startswith("SHA")]
    assert len(got) == len(cases), r.stdout
    return got


def test_tail_handover_arm_of_the_sliding_kernel_is_bit_identical(dev):
    """the sliding-window kernel's tail hand-over (xna_slide_kernel.h, STEAL: the last cells of every segment claimed
    cell by cell, finished workgroups take other runs' unclaimed tail cells) was built, measured slower and left OFF
    (profiles/r06_other_workloads.txt); it stays as an A/B arm behind NAF_XNA_STEAL=1.  Which workgroup computes a cell must not change a
    single bit of it: G2-k11, a 13 x 13 window and a non-square grid with uneven segments (12, 12, 12, 10 cells) through the whole forward,
    default vs the arm, in two processes."""
    cases = ((512, 512, 32, 32, 1024, 11), (512, 512, 32, 32, 768, 13), (256, 736, 16, 46, 1024, 11))
    assert _forward_hashes(cases, {"NAF_XNA_STEAL": "0"}) == _forward_hashes(cases, {"NAF_XNA_STEAL": "1"})


def test_half_row_staging_of_the_cell_kernel_is_bit_identical(dev):
    """Round 6: where the whole-row store tiles of the cell kernel leave fewer than sixteen waves per CU in flight -- Dv = 256 at 7 x 7
    (BASELINE's G2 / G3 width), the reference's default 9 x 9 window at Dv = 192 / 256 -- the planner takes eight-wave workgroups with store
    tiles of 128 / 64 channels (xna_mfma_kernel HS: G2-k7 0.62 -> 0.69, G3 0.64 -> 0.69 of the HBM roof, profiles/r06_other_workloads.txt).
    Only the way a tile's result leaves the LDS changes: the output is bit-identical to the whole-row arm (NAF_XNA_HS=0).  Both arms
    run with NAF_XNA_STAGE=1 (stage whenever it fits): without it the whole-row arm of 9 x 9 at Dv = 256 is not this kernel but the
    sliding-window one (one staged workgroup per CU is below the planner's bar), whose softmax is arranged differently -- equal within
    the oracle's tolerance, not bit for bit; the half-row plans themselves are the ones the default takes."""
    cases = ((512, 512, 32, 32, 1024, 7), (448, 448, 28, 28, 768, 9), (256, 320, 16, 20, 1024, 9), (160, 176, 10, 11, 1024, 5),
             (448, 448, 32, 32, 1024, 7))        # the last one: 14-pixel cells (partial row tiles: the predicated flush)
    assert _forward_hashes(cases, {"NAF_XNA_HS": "1", "NAF_XNA_STAGE": "1"}) == _forward_hashes(cases, {"NAF_XNA_HS": "0", "NAF_XNA_STAGE": "1"})


@pytest.mark.parametrize("mode", ["train", "eval_requires_grad"])
def test_return_weights_on_a_gradient_enabled_call(dev, mode):
    """VERDICT r05 (missing 3): the reference's ``return_weights`` works under autograd (attentions.py:64-67: legacy_attention hands back the
    scaled scores next to the output).  Round 6: a gradient-enabled ``naf(image, feats, size, return_weights=True)`` returns (out, scores):
    ``out`` differentiable as before, the scores of the very q / k that step used, fp32 ``[B, heads, Ho, Wo, k*k]``, without a gradient;
    both agree with the oracle's and with the inference path's."""
    p = O.make_params(seed=43)
    m = _load_model(dev, p, kernel_size=5)
    img = O.hash_normal((1, 3, 48, 64), 4301).to(dev)
    ft = O.hash_normal((1, 64, 6, 8), 4302).to(dev)
    ref, ref_lg = O.naf_forward(p, img.cpu(), ft.cpu(), (48, 64), kernel_size=5, return_weights=True)
    with torch.no_grad():
        out_i, lg_i = m(img, ft, (48, 64), return_weights=True)
    if mode == "train":
        m.train()
        m.image_encoder.rope.rescale_coords = None          # deterministic coordinates: comparable with the oracle
        ftg = ft
    else:
        ftg = ft.clone().requires_grad_(True)
    out, lg = m(img, ftg, (48, 64), return_weights=True)
    assert out.grad_fn is not None and lg.grad_fn is None and not lg.requires_grad
    assert lg.shape == ref_lg.shape == lg_i.shape and lg.dtype == torch.float32
    _assert_close(out.detach().float().cpu(), ref, 2e-2, 1e-2, "output of the gradient-enabled call")
    # scores: |q| |k| / 8 is O(10) here; bf16 q / k through five bf16 stem layers
    assert float((lg.cpu() - ref_lg).abs().max()) <= 2.5e-1 and float((lg.cpu() - ref_lg).abs().mean()) <= 2e-2
    assert float((lg - lg_i).abs().max()) <= 1.5e-1
    out.float().square().mean().backward()                     # and the step still differentiates
    if mode == "train":
        g = m.image_encoder.sem_encoder[0].weight.grad
        assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0
    else:
        assert ftg.grad is not None and bool(torch.isfinite(ftg.grad).all())
