"""NAF upsampler module: the reference's public operator (``naf(image, lr_features, target_size)``)
on MI355X-native kernels.

Mirrors the reference's module tree so that ``state_dict`` keys, constructor arguments and the
attributes callers read are identical (src/model/naf.py:72-116, src/layers/convolutions.py:6-92,
src/layers/rope.py:39-83, src/layers/attentions.py:32-47):

    NAF.image_encoder.encoder / .sem_encoder    nn.Sequential(Conv2d, EncBlock, ...)  (1x1 / 3x3-reflect)
    NAF.image_encoder.rope.periods              persistent buffer [D_head / 4]
    NAF.query_encoder, NAF.key_encoder          parameter-free
    NAF.upsampler.kernel_size / .num_heads / .scale / .dilation

The forward pass differs in HOW, not WHAT:
  * K/V are never upsampled to the output grid and the score tensor is never materialised
    (attentions.py:60-61,20-23 do both);
  * RoPE, the identity query encoder and the key pooling are one HIP pass (naf_rope_pool_fwd);
  * attention runs in the fused HIP kernel (naf_xna_fwd): MFMA cell kernel for integer ratios,
    table-driven kernel otherwise and for ``return_weights``.
Contract: bf16 queries/keys/values, fp32 softmax and accumulation; the output dtype follows
``features`` (bf16 -> bf16, anything else -> fp32) and is returned as a logical [B, C, Ho, Wo]
view of a channels-last buffer -- the same view the reference returns on its fused NATTEN path
(attentions.py:75).  Inputs must live on a ROCm device; there is no CPU path.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


class EncBlock(nn.Module):
    """GroupNorm(8) -> SiLU -> Conv -> GroupNorm(8) -> SiLU -> Conv, reflect padding, no skip
    (convolutions.py:6-64 with residual=False, in == out channels so no shortcut conv exists)."""

    def __init__(self, channels: int, kernel_size: int, num_groups: int = 8, bias: bool = True):
        super().__init__()
        pad = kernel_size // 2
        self.norm1 = nn.GroupNorm(num_groups, channels)
        self.conv1 = nn.Conv2d(channels, channels, kernel_size, padding=pad, padding_mode="reflect", bias=bias)
        self.norm2 = nn.GroupNorm(num_groups, channels)
        self.conv2 = nn.Conv2d(channels, channels, kernel_size, padding=pad, padding_mode="reflect", bias=bias)


def make_branch(in_dim: int, hidden: int, kernel_size: int, ks_res: int, num_layers: int) -> nn.Sequential:
    """Same parameter layout as the reference's ``encoder(...)`` (convolutions.py:67-92)."""
    return nn.Sequential(
        nn.Conv2d(in_dim, hidden, kernel_size, padding=kernel_size // 2, padding_mode="reflect", bias=True),
        *[EncBlock(hidden, ks_res) for _ in range(num_layers)],
    )


class RoPE(nn.Module):
    """Holds the `periods` buffer (rope.py:77-81,128-135) and caches the device sin/cos tables per
    output size, like the reference caches coordinates per (H, W) (rope.py:159-161).  ``tables`` are the deterministic
    eval-mode tables the HIP kernels read; ``train_tables`` adds the train-time coordinate augmentation (shift / jitter /
    rescale, rope.py:107-124) for ``NAF.forward_train`` when the module is in training mode."""

    def __init__(self, embed_dim: int, num_heads: int, base: float = 100.0, rescale_coords: Optional[float] = None):
        super().__init__()
        if embed_dim % (4 * num_heads):
            raise AssertionError("embed_dim must be divisible by 4 * num_heads")
        self.num_heads = num_heads
        self.base = base
        self.D_head = embed_dim // num_heads
        self.rescale_coords = rescale_coords
        self.shift_coords = None          # rope.py:49-51: not set by NAF (naf.py:29), kept for parity of the attribute set
        self.jitter_coords = None
        d = self.D_head
        periods = base ** (2 * torch.arange(d // 4, dtype=torch.float32) / (d // 2))
        self.register_buffer("periods", periods, persistent=True)
        self._tables = None
        self._tables_key = None
        # train-time coordinate augmentation (rope.py:107-124): the reference caches the coordinates per (H, W)
        # (rope.py:159-161), so its augmentation is drawn ONCE per resolution and even survives .eval(); the default here
        # redraws it on every training step (what the augmentation is for).  Set True to reproduce the reference's caching.
        self.cache_train_coords = False
        self._train_tables = {}

    def tables(self, Ho: int, Wo: int):
        p = self.periods
        key = (Ho, Wo, str(p.device), p.data_ptr(), p._version)
        if key != self._tables_key:
            self._tables = ops.rope_tables(p, Ho, Wo)
            self._tables_key = key
        return self._tables


def _rope_train_tables(rope: "RoPE", Ho: int, Wo: int):
    """cos / sin tables [Ho, 2, P], [Wo, 2, P] like ops.rope_tables, with the reference's train-time augmentation of the
    coordinates (rope.py:107-124): one uniform shift per axis, one log-uniform jitter per axis, one log-uniform rescale for
    both, drawn per call while ``rope.training``.  'separate' normalisation (rope.py:98-100)."""
    import math
    p = rope.periods
    dev = p.device
    cy = 2.0 * (torch.arange(Ho, device=dev, dtype=torch.float32) + 0.5) / Ho - 1.0
    cx = 2.0 * (torch.arange(Wo, device=dev, dtype=torch.float32) + 0.5) / Wo - 1.0
    if rope.training and rope.shift_coords is not None:
        sh = torch.empty(2, device=dev).uniform_(-rope.shift_coords, rope.shift_coords)
        cy, cx = cy + sh[0], cx + sh[1]
    if rope.training and rope.jitter_coords is not None:
        j = torch.empty(2, device=dev).uniform_(-math.log(rope.jitter_coords), math.log(rope.jitter_coords)).exp()
        cy, cx = cy * j[0], cx * j[1]
    if rope.training and rope.rescale_coords is not None:
        r = torch.empty(1, device=dev).uniform_(-math.log(rope.rescale_coords), math.log(rope.rescale_coords)).exp()
        cy, cx = cy * r, cx * r
    ay = (2.0 * math.pi * cy)[:, None] / p[None, :]                                   # rope.py:139
    ax = (2.0 * math.pi * cx)[:, None] / p[None, :]
    return torch.stack([ay.cos(), ay.sin()], dim=1), torch.stack([ax.cos(), ax.sin()], dim=1)


class _GroupNormTrain(torch.autograd.Function):
    """GroupNorm for the training stem.  ATen's ROCm forward computes the statistics with one workgroup per
    (sample, group) -- 8 workgroups at batch 1, 2.1 ms per layer at 448^2 (profiles/r01_train_step.txt).  Here the
    moments are a per-channel reduction over the channels-last rows ([HW, C]: C outputs, parallel over HW) folded
    into groups on [B, C] values; the normalisation is one fused multiply-add; the backward is ATen's own
    native_group_norm_backward (0.25 ms), fed with the saved mean / rstd."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        B, C, H, W = x.shape
        xv = x.permute(0, 2, 3, 1).reshape(B, H * W, C).float()                       # rows of C contiguous channels (a view
        HW = H * W                                                                    # for channels-last input), fp32 statistics
        S = next((d for d in (512, 256, 128, 64, 32, 16) if HW % d == 0 and HW // d >= 8), 0)
        if C >= 96 or S == 0:
            var_c, mean_c = torch.var_mean(xv, dim=1, unbiased=False)                 # [B, C], Welford
        else:
            # few channels = few outputs: ATen's column reduction then runs on a handful of workgroups (0.3 ms for 48
            # channels at 256^2); reduce in two stages, S partial sums per channel first -- mean, then CENTRED squares
            mean_c = xv.view(B, S, HW // S, C).sum(2).sum(1) / HW
            dc = xv - mean_c.unsqueeze(1)
            var_c = (dc * dc).view(B, S, HW // S, C).sum(2).sum(1) / HW
        # fold channels into groups without E[x^2] - mean^2 (activations with |mean| >> std would lose every variance
        # bit): var_g = mean_c(var_c) + mean_c((mean_c - mean_g)^2), both terms non-negative
        mean_cg = mean_c.view(B, groups, -1)
        mean = mean_cg.mean(-1)                                                        # [B, G]
        var = var_c.view(B, groups, -1).mean(-1) + ((mean_cg - mean.unsqueeze(-1)) ** 2).mean(-1)
        rstd = torch.rsqrt(var + eps)
        scale = (rstd.unsqueeze(-1) * weight.float().view(groups, -1)).reshape(B, C)
        shift = bias.float().unsqueeze(0) - (mean.unsqueeze(-1) * scale.view(B, groups, -1)).reshape(B, C)
        ctx.save_for_backward(x, mean, rstd, weight)
        ctx.groups = groups
        return torch.addcmul(shift.view(B, C, 1, 1).to(x.dtype), x, scale.view(B, C, 1, 1).to(x.dtype))

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, weight = ctx.saved_tensors
        B, C, H, W = x.shape
        # ATen's kernel indexes dense NCHW memory (autograd hands it grad.contiguous() as well)
        # (and one dtype: under autocast x / dy arrive in bf16 while the statistics and the affine parameters are fp32)
        dx, dw, db = torch.ops.aten.native_group_norm_backward(dy.float().contiguous(), x.float().contiguous(), mean, rstd,
                                                               weight.float(), B, C, H * W, ctx.groups, [True, True, True])
        return dx.to(x.dtype).contiguous(memory_format=torch.channels_last), dw.to(weight.dtype), db.to(weight.dtype), None, None


class _ReflectPad(torch.autograd.Function):
    """F.pad(mode="reflect") whose backward folds the border strips back with slice adds (ATen's
    reflection_pad2d_backward is a scalar atomic kernel: 0.85 ms per layer at 448^2 on gfx950)."""

    @staticmethod
    def forward(ctx, x, pad):
        ctx.pad = pad
        return F.pad(x, (pad, pad, pad, pad), mode="reflect")

    @staticmethod
    def backward(ctx, g):
        p = ctx.pad
        H, W = g.shape[-2] - 2 * p, g.shape[-1] - 2 * p
        gx = g[..., :, p:p + W].clone()                                   # columns first: [.., H + 2p, W]
        gx[..., :, 1:p + 1] += g[..., :, :p].flip(-1)                     # left strip mirrors columns 1..p
        gx[..., :, W - 1 - p:W - 1] += g[..., :, p + W:].flip(-1)         # right strip mirrors columns W-1-p..W-2
        out = gx[..., p:p + H, :].clone()
        out[..., 1:p + 1, :] += gx[..., :p, :].flip(-2)
        out[..., H - 1 - p:H - 1, :] += gx[..., p + H:, :].flip(-2)
        return out, None


def _stem_layers(seq: nn.Sequential):
    """(norm, conv) pairs of a branch after its first convolution (convolutions.py:52-61: norm1 -> SiLU -> conv1, norm2 -> ...)."""
    out = []
    for blk in list(seq)[1:]:
        out += [(blk.norm1, blk.conv1), (blk.norm2, blk.conv2)]
    return out


class _HipStem(torch.autograd.Function):
    """Both branches of the conv stem (convolutions.py:67-92) as ONE differentiable op on the HIP kernels: the forward is the
    inference stem (naf_stem_conv0_fwd / naf_stem_conv_fwd, bf16 activations, fp32 accumulation, fp64 GroupNorm sums) with
    every layer's input kept; the backward walks the layers down with
      data gradient    naf_stem_conv_fwd in plain mode on the flipped / transposed weights (3x3: over the output gradient in a
                       2-pixel zero border, i.e. on the reflect-PADDED domain),
      norm + SiLU      naf_stem_act_bwd (folds the padded border back on load, returns the GroupNorm affine gradients),
      weight gradient  naf_stem_wgrad: a GEMM contracted over pixels (transposing LDS reads feed both MFMA operands), with
                       a = SiLU(GroupNorm(x)) recomputed in its loader.
    bf16 roundings of stored activations / gradients are treated as identities (as torch.autocast does for bf16 convolutions:
    this is the reference's use_bf16 training mode, train.py:120).  Every width the forward stem serves (round 6: hidden widths that
    are multiples of 16 up to 256 -- the reference's denoising models, denoising.py:209-220 -- through stem_generic.hip's plain
    convolution, stem_generic_bwd.hip's weight gradient and the width-general norm / first-convolution kernels of stem_bwd.hip)."""

    @staticmethod
    def forward(ctx, enc, image, *params):
        B, _, H, W = image.shape
        dev = image.device
        branches = (enc.encoder, enc.sem_encoder)
        hid = enc.encoder[0].out_channels
        nlayer = len(_stem_layers(enc.encoder))
        stats = ops.new_stats(B, dev, lead=(2, nlayer + 1))
        cat = torch.empty((B, H, W, 2 * hid), dtype=torch.bfloat16, device=dev)
        img = image.detach()
        if img.dtype not in (torch.float32, torch.bfloat16):
            img = img.float()
        saved = []
        for br, seq in enumerate(branches):
            conv0 = seq[0]
            ys = [torch.empty((B, H, W, hid), dtype=torch.bfloat16, device=dev) for _ in range(max(nlayer, 1))]
            dst = ys[0] if nlayer else cat[..., br * hid:(br + 1) * hid]
            ops.stem_conv0(img, conv0.weight.detach().float().contiguous(), conv0.bias.detach().float(), dst, stats[br, 0])
            for li, (norm, conv) in enumerate(_stem_layers(seq)):
                last = li == nlayer - 1
                dst = cat[..., br * hid:(br + 1) * hid] if last else ys[li + 1]
                ops.stem_conv(ys[li], stats[br, li], norm.weight.detach().float(), norm.bias.detach().float(), norm.eps,
                              enc._packed(conv), conv.bias.detach().float(), dst, None if last else stats[br, li + 1])
            saved.append(ys)
        # saved through autograd's own slots (version-counter checks, freed with the graph); the module is only consulted for its
        # layer structure and current parameter values
        ctx.save_for_backward(image, stats, *[y for ys in saved for y in ys])
        ctx.enc, ctx.nsaved, ctx.nparams = enc, [len(ys) for ys in saved], len(params)
        return cat.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        enc = ctx.enc
        image, stats, *flat = ctx.saved_tensors
        saved, pos = [], 0
        for n in ctx.nsaved:
            saved.append(flat[pos:pos + n])
            pos += n
        B, _, H, W = image.shape
        dev = g.device
        hid = enc.encoder[0].out_channels
        gcl = g.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()          # [B, H, W, 2 hid]; a no-op for a channels-last gradient
        grads = []
        dimage = None
        need_img = ctx.needs_input_grad[1]
        # every accumulator of the step -- weight / bias gradients (fp32 atomics) and the norm layers' sums (fp64 atomics) -- lives in ONE zeroed
        # buffer per dtype, and the sums are reduced over the batch once at the end (a memset, a reduction and two casts per LAYER before)
        branches = (enc.encoder, enc.sem_encoder)
        sizes = []
        for seq in branches:
            lay = _stem_layers(seq)
            kk, k0 = (lay[0][1].kernel_size[0] if lay else 1), seq[0].kernel_size[0]
            sizes.append([(3 * k0 * k0 + 1) * hid] + [kk * kk * hid * hid + hid] * len(lay))
        acc32 = torch.zeros((sum(sum(z) for z in sizes),), dtype=torch.float32, device=dev)
        nl_max = max(len(z) - 1 for z in sizes)
        acc64 = torch.zeros((2, max(nl_max, 1), B, hid, 2), dtype=torch.float64, device=dev)
        pos32 = 0
        for br, seq in enumerate(branches):
            ys = saved[br]
            layers = _stem_layers(seq)
            k = layers[0][1].kernel_size[0] if layers else 1
            out0 = acc32[pos32:pos32 + sizes[br][0]]
            pos32 += sizes[br][0]
            outs = []
            for z in sizes[br][1:]:
                outs.append(acc32[pos32:pos32 + z])
                pos32 += z
            # 3x3 branch: a layer's output gradient lives in the interior of a buffer with a 2-pixel ZERO border (what the
            # data-gradient convolution of the padded domain reads); naf_stem_act_bwd writes the next one straight into
            # the other buffer's interior, so only the branch's incoming gradient is ever copied
            ext = [torch.zeros((B, H + 4, W + 4, hid), dtype=torch.bfloat16, device=dev) for _ in range(2)] if k == 3 else None
            if k == 3:
                gl = ext[0][:, 2:H + 2, 2:W + 2]
                gl.copy_(gcl[..., br * hid:(br + 1) * hid])
            else:
                gl = gcl[..., br * hid:(br + 1) * hid]
            bgrads = []
            for li in range(len(layers) - 1, -1, -1):
                norm, conv = layers[li]
                w = conv.weight.detach()
                gw, gb = norm.weight.detach().float(), norm.bias.detach().float()
                # weight / bias gradient: naf_stem_wgrad (pixel-contraction GEMM; a = SiLU(GroupNorm(x)) recomputed in its loader)
                dw, db = ops.stem_wgrad(gl, ys[li], stats[br, li], gw, gb, norm.eps, k, with_bias=True, out=outs[li])
                # data gradient: the same conv kernel, plain, on the flipped / transposed weights
                wt = ops.pack_conv_weight(w.flip(2, 3).transpose(0, 1))
                if k == 3:
                    cur = (len(layers) - 1 - li) & 1
                    full = torch.empty_like(ext[cur])
                    ops.stem_conv_plain(ext[cur], wt, full)
                    dx = ext[cur ^ 1][:, 2:H + 2, 2:W + 2]
                    ops.stem_act_bwd(full[:, 1:H + 3, 1:W + 3], ys[li], stats[br, li], gw, gb, norm.eps, dx, fold=True, sums=acc64[br, li])
                    del full
                else:
                    da = torch.empty((B, H, W, hid), dtype=torch.bfloat16, device=dev)
                    ops.stem_conv_plain(gl, wt, da)
                    dx = torch.empty((B, H, W, hid), dtype=torch.bfloat16, device=dev)
                    ops.stem_act_bwd(da, ys[li], stats[br, li], gw, gb, norm.eps, dx, fold=False, sums=acc64[br, li])
                    del da
                bgrads.append((li, norm, dw.to(conv.weight.dtype), db.to(conv.bias.dtype)))
                gl = dx
            # first convolution (3 -> hid on the image): naf_stem_conv0_wgrad for the parameters and, when the image wants a
            # gradient, naf_stem_conv0_dgrad (round 6: no ATen / MIOpen convolution is left on the training path)
            conv0 = seq[0]
            dw0, db0 = ops.stem_conv0_wgrad(gl, image.detach(), conv0.kernel_size[0], out=out0)
            grads.append((br, dw0.to(conv0.weight.dtype), db0.to(conv0.bias.dtype), bgrads[::-1]))
            if need_img:
                if dimage is None:
                    dimage = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
                    ops.stem_conv0_dgrad(gl, conv0.weight.detach().float().contiguous(), dimage, accumulate=False)
                else:
                    ops.stem_conv0_dgrad(gl, conv0.weight.detach().float().contiguous(), dimage, accumulate=True)
        s32 = acc64.sum(2).float()      # [branch, layer, hid, {d bias, d weight}]: one reduction over the batch for every norm layer
        flat = []
        for br, dw0, db0, per_layer in grads:          # the order of _hip_stem_params
            flat += [dw0, db0]
            for li, norm, dw, db in per_layer:
                flat += [s32[br, li, :, 1].to(norm.weight.dtype), s32[br, li, :, 0].to(norm.bias.dtype), dw, db]
        grads = flat
        assert len(grads) == ctx.nparams
        return (None, dimage.to(image.dtype) if dimage is not None else None, *grads)


def _hip_stem_params(enc):
    """Parameters in the order _HipStem.backward returns their gradients."""
    out = []
    for seq in (enc.encoder, enc.sem_encoder):
        out += [seq[0].weight, seq[0].bias]
        for norm, conv in _stem_layers(seq):
            out += [norm.weight, norm.bias, conv.weight, conv.bias]
    return out


class ImageEncoder(nn.Module):
    """Guidance encoder (naf.py:11-52).  Both conv branches run through the library's fused HIP stem
    (naf_stem_conv0_fwd / naf_stem_conv_fwd: GroupNorm+SiLU+conv in one MFMA pass per layer): the hand-scheduled
    weight-stationary kernels at the default width (dim 256 -> 128 hidden channels), the general kernels of
    stem_generic.hip at any other width that is a multiple of 16 up to 256 (the reference's denoising models, dim 96 ... 512).
    ``stem_impl = "torch"`` (MIOpen ops in ``stem_dtype``) is an explicit A/B switch, never a silent fallback.
    RoPE is applied by the caller's fused rope+pool kernel."""

    def __init__(self, in_channels=3, out_channels=256, heads_rope=1, use_encoder=True, rope_base=None,
                 rope_rescale=None, img_layers=2):
        super().__init__()
        self.use_encoder = use_encoder
        self.out_channels = out_channels
        self.encoder = make_branch(in_channels, out_channels // 2, 1, 1, img_layers)
        self.sem_encoder = make_branch(in_channels, out_channels // 2, 3, 3, img_layers)
        self.rope = RoPE(out_channels, num_heads=heads_rope, base=rope_base, rescale_coords=rope_rescale)
        self.stem_dtype = torch.bfloat16
        self.stem_impl = "hip"      # "hip": fused HIP stem; "torch": MIOpen ops (A/B only)
        self.fuse_conv0 = True      # 1x1 branch: first block layer recomputes conv0 instead of reading it

    @staticmethod
    def _conv(x, conv: nn.Conv2d, dt):
        pad = conv.kernel_size[0] // 2
        if pad:
            x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
        return F.conv2d(x, conv.weight.to(dt), None if conv.bias is None else conv.bias.to(dt))

    def _branch(self, x, seq: nn.Sequential, dt):
        x = self._conv(x, seq[0], dt)
        for blk in list(seq)[1:]:
            # (_GroupNormTrain: ATen's group_norm forward runs its statistics on B * groups workgroups, see its docstring)
            x = F.silu(_GroupNormTrain.apply(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.num_groups, blk.norm1.eps))
            x = self._conv(x, blk.conv1, dt)
            x = F.silu(_GroupNormTrain.apply(x, blk.norm2.weight, blk.norm2.bias, blk.norm2.num_groups, blk.norm2.eps))
            x = self._conv(x, blk.conv2, dt)
        return x

    # ---- differentiable torch stem of forward_train ---------------------------------------------------
    # Same math as _branch (convolutions.py:52-92) with the three ops whose ATen ROCm kernels dominate a training
    # step at batch 1 replaced by equivalent compositions (profiles/r01_train_step.txt): GroupNorm statistics
    # (ATen launches one workgroup per (sample, group) = 8 workgroups: 2.1 ms per layer at 448^2), the reflect-pad
    # backward (0.85 ms per layer) and -- in forward_train -- the adaptive-average-pool backward (atomics, 2.6 ms).
    @staticmethod
    def _group_norm_train(x, norm: nn.GroupNorm):
        return _GroupNormTrain.apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps)

    def _conv_train(self, x, conv: nn.Conv2d):
        pad = conv.kernel_size[0] // 2
        if pad:
            x = _ReflectPad.apply(x, pad)
        return F.conv2d(x, conv.weight, conv.bias)

    def _branch_train(self, x, seq: nn.Sequential):
        x = self._conv_train(x, seq[0])
        for blk in list(seq)[1:]:
            x = self._conv_train(F.silu(self._group_norm_train(x, blk.norm1)), blk.conv1)
            x = self._conv_train(F.silu(self._group_norm_train(x, blk.norm2)), blk.conv2)
        return x

    # ---- fused HIP stem (hidden width a multiple of 16 up to 256, GroupNorm(8)) --------------------
    def _hip_stem_ok(self) -> bool:
        c0 = self.encoder[0]
        return (self.use_encoder and c0.out_channels % 16 == 0 and 16 <= c0.out_channels <= 256 and c0.in_channels == 3
                and all(b.norm1.num_groups == 8 for b in list(self.encoder)[1:]))

    def _hip_stem_default_width(self) -> bool:
        return self._hip_stem_ok() and self.encoder[0].out_channels == 128

    def _hip_train_stem_ok(self) -> bool:
        """Widths ``_HipStem`` (forward with saved activations + HIP backward kernels) serves: all the forward stem does."""
        return self._hip_stem_ok()

    def _packed(self, conv: nn.Conv2d) -> torch.Tensor:
        """``ops.pack_conv_weight`` of a conv weight (bf16, the order naf_stem_conv_fwd reads), cached until the parameter changes."""
        cache = self.__dict__.setdefault("_wcache", {})
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            k = w.shape[-1]
            hit = (key, ops.pack_conv_weight(w))
            cache[id(conv)] = hit
        return hit[1]

    def _stem_hip(self, image: torch.Tensor) -> torch.Tensor:
        """Both conv branches through naf_stem_conv0_fwd / naf_stem_conv_fwd; returns the concatenated
        guidance as a logical [B, 256, H, W] view of a channels-last bf16 buffer (naf.py:31-33)."""
        B, _, H, W = image.shape
        dev = image.device
        branches = (self.encoder, self.sem_encoder)
        hid = self.encoder[0].out_channels
        nstage = 1 + 2 * (len(self.encoder) - 1)
        stats = ops.new_stats(B, dev, lead=(2, nstage))      # one memset for all sums
        cat = torch.empty((B, H, W, 2 * hid), dtype=torch.bfloat16, device=dev)
        bufs = [torch.empty((B, H, W, hid), dtype=torch.bfloat16, device=dev) for _ in range(2)]
        for br, seq in enumerate(branches):
            conv0 = seq[0]
            w0, b0 = conv0.weight.detach().float().contiguous(), conv0.bias.detach().float()
            # 1x1 branch: the conv0 activation is never stored -- one statistics-only pass, then the first
            # GroupNorm/SiLU/conv layer recomputes it from the image (bit-identical, 0.5 GB less traffic)
            recompute = (self.fuse_conv0 and hid == 128 and conv0.kernel_size[0] == 1 and nstage > 1
                         and seq[1].conv1.kernel_size[0] == 1)
            last = nstage == 1
            dst = cat[..., br * hid:(br + 1) * hid] if last else bufs[0]
            ops.stem_conv0(image, w0, b0, None if recompute else dst, stats[br, 0])
            cur, st = dst, 0
            for blk in list(seq)[1:]:
                for norm, conv in ((blk.norm1, blk.conv1), (blk.norm2, blk.conv2)):
                    st += 1
                    last = st == nstage - 1
                    dst = cat[..., br * hid:(br + 1) * hid] if last else bufs[st % 2]
                    ops.stem_conv(None if (recompute and st == 1) else cur, stats[br, st - 1], norm.weight.detach().float(),
                                  norm.bias.detach().float(), norm.eps, self._packed(conv), conv.bias.detach().float(), dst,
                                  None if last else stats[br, st], first=(image, w0, b0) if (recompute and st == 1) else None)
                    cur = dst
        return cat.permute(0, 3, 1, 2)

    def guidance(self, image: torch.Tensor, output_size: Tuple[int, int]) -> torch.Tensor:
        """Pre-RoPE guidance features, logical [B, dim, Ho, Wo] (naf.py:37-49 + :31-35)."""
        ho, wo = output_size
        x = image
        if x.shape[-2] > 4 * ho or x.shape[-1] > 4 * wo:                       # naf.py:39-48
            size = (min(x.shape[-2], 4 * ho, 4 * wo), min(x.shape[-1], 4 * wo, 4 * ho))
            if x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.dtype in (torch.float32, torch.bfloat16):
                x = ops.preshrink_image(x, size)                               # the kernel naf_forward uses as well
            else:
                x = F.interpolate(x.float(), size=size, mode="bilinear", align_corners=False)
        if self.use_encoder and self.stem_impl == "hip":
            if not self._hip_stem_ok():
                raise RuntimeError(f"naf_amd: the HIP conv stem serves hidden widths that are multiples of 16 up to 256 with "
                                   f"GroupNorm(8) (got {self.encoder[0].out_channels}); there is no silent torch fallback -- set "
                                   f"image_encoder.stem_impl = 'torch' explicitly to run MIOpen ops")
            x = self._stem_hip(x)
        elif self.use_encoder:
            dt = self.stem_dtype
            x = x.to(dt).contiguous(memory_format=torch.channels_last)
            x = torch.cat([self._branch(x, self.encoder, dt), self._branch(x, self.sem_encoder, dt)], dim=1)
        if x.shape[-2:] != (ho, wo):                                           # naf.py:34
            if (x.dtype == torch.bfloat16 and x.is_cuda and x.shape[1] % 8 == 0
                    and x.is_contiguous(memory_format=torch.channels_last)):
                x = ops.pool_guidance(x, (ho, wo))                             # the kernel naf_forward uses as well
            else:
                x = F.adaptive_avg_pool2d(x, output_size=(ho, wo))
        return x.contiguous(memory_format=torch.channels_last)


class QueryEncoder(nn.Module):
    """Identity (naf.py:55-60)."""

    def forward(self, x):
        return x


class KeyEncoder(nn.Module):
    """``adaptive_avg_pool2d`` of the RoPE'd guidance to the feature grid (naf.py:63-69).  ``NAF.forward`` never calls it
    -- the pooling is fused into naf_rope_pool_fwd / naf_forward -- but callers that use the sub-module directly (as the
    reference allows) get the same result: HIP pooling kernel for channels-last bf16 device tensors, ATen otherwise."""

    def forward(self, x, features):
        size = tuple(int(v) for v in features.shape[-2:])
        if (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0
                and x.is_contiguous(memory_format=torch.channels_last)):
            return ops.pool_guidance(x, size)
        return F.adaptive_avg_pool2d(x, output_size=size)


class CrossAttention(nn.Module):
    """Cross-scale neighbourhood attention (attentions.py:32-75) on naf_xna_fwd."""

    def __init__(self, dim, num_heads, kernel_size=(9, 9), **kwargs):
        super().__init__()
        assert dim % num_heads == 0, "dim must be divisible by num_heads"
        self.num_heads = num_heads
        self.kernel_size = tuple(kernel_size)
        self.scale = (dim // num_heads) ** -0.5
        self.dilation = None

    def forward(self, q5, k5, values, return_weights=False, path="auto", rope_tables=None):
        """q5/k5: 5-D bf16 views from ops.rope_pool; values: [B, C, h, w] features.  With ``rope_tables`` q5 is
        the un-rotated guidance and the attention kernel rotates it on load."""
        B, C = values.shape[:2]
        if C % self.num_heads:
            raise ValueError(f"feature channels {C} not divisible by {self.num_heads} heads")   # einops error in the reference
        Ho, Wo = q5.shape[2:4]
        h, w = values.shape[-2:]
        self.dilation = (Ho // h, Wo // w)                                     # attentions.py:56-57
        vp = ops.pack_values(values)
        v5 = vp.view(B, h, w, self.num_heads, C // self.num_heads).permute(0, 3, 1, 2, 4)
        out_dtype = torch.bfloat16 if values.dtype == torch.bfloat16 else torch.float32
        res = ops.xna_forward(q5, k5, v5, self.kernel_size, out_dtype=out_dtype, return_logits=return_weights,
                              path=path, scale=self.scale, rope_tables=rope_tables)
        out5, logits = res if return_weights else (res, None)
        # [B, heads, Ho, Wo, Dv] view of a channels-last buffer -> logical [B, C, Ho, Wo]
        out = out5.permute(0, 2, 3, 1, 4).reshape(B, Ho, Wo, C).permute(0, 3, 1, 2)
        if values.dtype not in (torch.bfloat16, torch.float32):
            out = out.to(values.dtype)
        return (out, logits) if return_weights else out


def _walk_parameters(module: nn.Module):
    """Every parameter below ``module``, read fresh from the module tree (a re-assigned parameter or a replaced sub-module is seen),
    without ``Module.parameters()``'s names and de-duplication: the plan-cache key of every forward call (100 -> 25 us of host time)."""
    for p in module._parameters.values():
        if p is not None:
            yield p
    for child in module._modules.values():
        if child is not None:
            yield from _walk_parameters(child)


class GraphedForward:
    """One NAF forward captured in a hipGraph (``torch.cuda.CUDAGraph``) and replayed: the ~13 kernel launches of a
    forward become one graph launch, which removes the host-side launch gaps (0.06 ms of a 2.5 ms G1 step, 10 % of
    a 448^2 step).  Shapes are frozen at capture; ``__call__`` copies new inputs into the captured buffers (or, with no
    arguments, replays on whatever ``.image`` / ``.features`` hold -- write into them in place to skip the copies).
    The returned tensor is the graph's own output buffer: it is overwritten by the next replay."""

    def __init__(self, model: "NAF", image: torch.Tensor, features: torch.Tensor, output_size, warmup: int = 2,
                 capture_error_mode: str = "global"):
        if not (image.is_cuda and features.is_cuda):
            raise RuntimeError("GraphedForward needs device tensors")
        self.model, self.output_size = model, (int(output_size[0]), int(output_size[1]))
        self.image, self.features = image.clone(), features.clone()
        cur = torch.cuda.current_stream(image.device)
        side = torch.cuda.Stream(device=image.device)
        side.wait_stream(cur)
        # Always the fused inference forward, whatever mode the module is in: ``hubconf.naf()`` hands out a train-mode
        # module like the reference, and ``model(...)`` would then dispatch to ``forward_train`` (torch stem with saved
        # activations, a random RoPE jitter draw and an output with a grad_fn frozen into the graph).
        with torch.no_grad():
            with torch.cuda.stream(side):                   # warm-up outside the capture: caches, lazy module init
                for _ in range(max(1, warmup)):
                    model._forward_inference(self.image, self.features, self.output_size)
            cur.wait_stream(side)
            # the warm-up's workspace (1 GB at 1024^2) belonged to the throw-away stream: drop it before the capture allocates the
            # graph's own, instead of leaving it in the plan's pool until four other streams evict it
            side.synchronize()
            warm_plan = (model.__dict__.get("_plan_cache") or (None, None))[1]
            if warm_plan is not None:
                warm_plan.release_stream(image.device.index, int(side.cuda_stream))   # also drops the plan's "last used" reference to it
            # the capture runs on a stream of ours, and the second stream the forward forks onto (ops.forward_aux: the host lends
            # it to naf_forward_ex) is created BEFORE the capture begins
            cap = torch.cuda.Stream(device=image.device)
            ops.forward_aux(image.device, cap)
            self.graph = torch.cuda.CUDAGraph()
            # capture_error_mode="thread_local": other host threads may keep issuing (and synchronising with) eager work while this
            # one captures -- e.g. forwards of another module on another stream; their second streams are their own (ops.forward_aux)
            with torch.cuda.graph(self.graph, stream=cap, capture_error_mode=capture_error_mode):
                self.out = model._forward_inference(self.image, self.features, self.output_size)
        # The captured launches dereference device memory the graph does not own: the forward plan's workspace and
        # argument block, the RoPE tables, the packed bf16 weights.  They live in single-slot caches of the model that a
        # later eager call with other shapes (or a parameter update) replaces; hold them here so that a replay never
        # reads freed or recycled memory.
        enc = model.image_encoder
        plan_slot = model.__dict__.get("_plan_cache")
        plan = plan_slot[1] if plan_slot else None
        self._keep = [plan_slot, plan, getattr(plan, "_ws", None), getattr(plan, "_keep", None), enc.rope._tables,
                      dict(enc.__dict__.get("_wcache", {}))]

    def __call__(self, image: Optional[torch.Tensor] = None, features: Optional[torch.Tensor] = None) -> torch.Tensor:
        if image is not None:
            self.image.copy_(image)
        if features is not None:
            self.features.copy_(features)
        self.graph.replay()
        return self.out


class NAF(nn.Module):
    """Drop-in for the reference's ``NAF`` (naf.py:72-116): same constructor, same ``state_dict``."""

    def __init__(self, dim=256, heads_attn=4, heads_rope=4, kernel_size=9, use_encoder=True, rope_base=100.0,
                 rope_rescale=2.0, img_layers=2, **kwargs):
        super().__init__()
        self.image_encoder = ImageEncoder(in_channels=3, out_channels=dim, heads_rope=heads_rope,
                                          use_encoder=use_encoder, rope_base=rope_base, img_layers=img_layers,
                                          rope_rescale=rope_rescale)
        self.query_encoder = QueryEncoder()
        self.key_encoder = KeyEncoder()
        self.upsampler = CrossAttention(dim=dim, num_heads=heads_attn, kernel_size=(kernel_size, kernel_size))
        self.xna_path = "auto"
        self.fuse_rope = True      # rotate queries inside the attention kernel when the shapes allow it
        self.single_call = True    # issue the whole forward through naf_forward (one foreign call) when possible

    def guidance_qk(self, image, lr_size, output_size, fuse_for=None):
        """bf16 queries and pooled RoPE'd keys (5-D views) for ``image``.  Returns (q5, k5, rope_tables):
        rope_tables is None when q5 is already rotated, or the (tab_y, tab_x) pair when q5 is the un-rotated
        guidance that the attention kernel rotates on load (``fuse_for`` = (Dv, out_dtype) of the attention call
        asks for that; it is granted when the shapes allow it, see ops.xna_rope_fusable)."""
        ho, wo = int(output_size[0]), int(output_size[1])
        enc = self.image_encoder
        with ops._Timed("stem"):
            x = enc.guidance(image, (ho, wo))
        tab_y, tab_x = enc.rope.tables(ho, wo)
        heads_rope, heads_attn = enc.rope.num_heads, self.upsampler.num_heads
        same = heads_attn == heads_rope
        if fuse_for is not None and self.fuse_rope and same and x.dtype == torch.bfloat16:
            B, Cq = x.shape[:2]
            xq = x.permute(0, 2, 3, 1)                                            # [B, Ho, Wo, C] view
            if xq.stride(3) == 1:
                q5 = xq.unflatten(3, (heads_rope, Cq // heads_rope)).permute(0, 3, 1, 2, 4)
                if ops.xna_rope_fusable(q5, lr_size, fuse_for[0], self.upsampler.kernel_size, (tab_y, tab_x),
                                        out_dtype=fuse_for[1], path=self.xna_path):
                    _, k5 = ops.rope_pool(x, tab_y, tab_x, heads_rope, lr_size, write_q=False)
                    return q5, k5, (tab_y, tab_x)
        q5, k5 = ops.rope_pool(x, tab_y, tab_x, heads_rope, lr_size, q_layout="head_major" if same else "channels_last")
        if not same:        # re-split the channel axis for attention: pure views of channels-last buffers
            B, _, Ho, Wo, _ = q5.shape
            dim = enc.out_channels
            q5 = q5.permute(0, 2, 3, 1, 4).reshape(B, Ho, Wo, heads_attn, dim // heads_attn).permute(0, 3, 1, 2, 4)
            h, w = k5.shape[2:4]
            k5 = k5.permute(0, 2, 3, 1, 4).reshape(B, h, w, heads_attn, dim // heads_attn).permute(0, 3, 1, 2, 4)
        return q5, k5, None

    def _forward_plan(self, image, features, output_size):
        """``ops.ForwardPlan`` (one ``naf_forward`` call per forward) when the configuration allows it, else None:
        default width through the fused stem (any image / output / feature sizes, either head split, return_weights).
        Cached until a parameter, the shapes, strides or dtypes change."""
        enc = self.image_encoder
        ho, wo = int(output_size[0]), int(output_size[1])
        if not (enc.use_encoder and enc.stem_impl == "hip" and enc._hip_stem_default_width() and enc.fuse_conv0 and self.fuse_rope):
            return None
        if image.shape[1] != 3 or self.xna_path != "auto":
            return None
        if features.dtype not in (torch.bfloat16, torch.float32):
            return None
        if image.dtype not in (torch.bfloat16, torch.float32) or features.shape[1] % self.upsampler.num_heads:
            return None
        if self.upsampler.kernel_size[0] != self.upsampler.kernel_size[1]:
            return None
        prm_key = tuple((p.data_ptr(), p._version) for p in _walk_parameters(self)) + (enc.rope.periods.data_ptr(), enc.rope.periods._version)
        key = (prm_key, tuple(image.shape), tuple(image.stride()), image.dtype, tuple(features.shape), tuple(features.stride()),
               features.dtype, str(image.device), (ho, wo))
        hit = self.__dict__.get("_plan_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        branches = []
        for seq in (enc.encoder, enc.sem_encoder):
            c0 = seq[0]
            layers = [(n.weight.detach().float().contiguous(), n.bias.detach().float().contiguous(), enc._packed(c),
                       c.bias.detach().float().contiguous())
                      for blk in list(seq)[1:] for n, c in ((blk.norm1, blk.conv1), (blk.norm2, blk.conv2))]
            if len(layers) > 8 or any(c.kernel_size[0] != seq[1].conv1.kernel_size[0] for blk in list(seq)[1:] for c in (blk.conv1, blk.conv2)):
                return None
            branches.append((c0.weight.detach().float().contiguous(), c0.bias.detach().float().contiguous(), c0.kernel_size[0],
                             seq[1].conv1.kernel_size[0] if len(seq) > 1 else 1, layers))
        if not branches[0][4]:
            return None
        eps = enc.encoder[1].norm1.eps
        plan = ops.ForwardPlan(branches, len(branches[0][4]), eps, enc.rope.tables(ho, wo), image, features,
                               self.upsampler.num_heads, self.upsampler.kernel_size[0],
                               torch.bfloat16 if features.dtype == torch.bfloat16 else torch.float32, self.upsampler.scale,
                               output_size=(ho, wo), heads_rope=enc.rope.num_heads)
        plan = plan if plan.supported else None
        self.__dict__["_plan_cache"] = (key, plan)
        return plan

    def capture(self, image, features, output_size, capture_error_mode: str = "global") -> GraphedForward:
        """Capture this forward for the given shapes in a hipGraph; see ``GraphedForward``."""
        return GraphedForward(self, image, features, output_size, capture_error_mode=capture_error_mode)

    def forward_train(self, image, features, output_size, amp="auto", return_weights=False):
        """Differentiable forward for training (train.py:127-137): gradients reach the encoder parameters, the image
        and the features.  The attention and its backward are the HIP kernels (naf_xna_fwd / naf_xna_bwd through
        ``ops.XnaFunction``); the conv stem, RoPE and key pooling run as torch ops so that autograd can
        differentiate them (the fused inference stem has no backward).  In ``.train()`` mode the RoPE coordinates get the
        reference's random rescale (rope.py:107-124, NAF's rope_rescale); in ``.eval()`` mode they are deterministic.  Every geometry
        has a backward kernel (``ops.xna_backward_select``): the MFMA cell kernel (integer ratio, Wo/w a multiple of 16, window <= 13
        with K/V windows inside the LDS), the row-streaming matrix-core kernel (every other integer ratio: the reference's own training
        geometry 16^2 -> 32^2, patch-14 backbones, the denoising call), the table-driven scalar kernel for the rest.
        ``return_weights``: also returns the scaled pre-softmax scores [B, heads, Ho, Wo, k*k] of the q / k this step used, fp32, without a
        gradient (the reference's ``return_weights`` under autograd, attentions.py:64-67; its callers only display them).
        ``amp="auto"`` (the default, and what ``model(image, feats, size)`` uses when a gradient is wanted; round 6) trains through the
        library's own differentiable stem ``_HipStem`` whenever ``image_encoder.stem_impl == "hip"`` and the width has HIP training
        kernels -- with or without ``torch.autocast``: its contract (bf16 activations between layers, fp32 accumulation, fp64 GroupNorm
        sums) is what the inference forward computes, so training and inference see the same function.  The torch stem stays as the
        explicit A/B arm: ``amp=False`` = fp32 MIOpen convolutions, ``amp=True`` = the reference's ``use_bf16`` mode (bf16 stem
        convolutions under ``torch.autocast``, train.py:120, denoising.py:209; GroupNorm statistics, RoPE and pooling fp32);
        ``stem_impl = "torch"`` makes "auto" choose between those two by the ambient autocast state.  ``amp="hip"`` insists on
        ``_HipStem`` (raises when the width is not served)."""
        if not (image.is_cuda and features.is_cuda):
            raise RuntimeError("naf_amd.NAF runs only on a ROCm device (HIP kernels, no CPU fallback)")
        enc = self.image_encoder
        ho, wo = int(output_size[0]), int(output_size[1])
        h, w = features.shape[-2:]
        heads_rope, heads = enc.rope.num_heads, self.upsampler.num_heads
        x = image
        if x.shape[-2] > 4 * ho or x.shape[-1] > 4 * wo:                       # naf.py:39-48
            x = F.interpolate(x.float(), size=(min(x.shape[-2], 4 * ho, 4 * wo), min(x.shape[-1], 4 * wo, 4 * ho)),
                              mode="bilinear", align_corners=False)
        if amp == "auto":
            if enc.use_encoder and enc.stem_impl == "hip" and enc._hip_train_stem_ok():
                amp = "hip"
            else:
                amp = bool(torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        if enc.use_encoder and amp == "hip":
            if not enc._hip_train_stem_ok():
                raise RuntimeError("forward_train(amp='hip'): the differentiable HIP stem serves hidden widths that are multiples of 16 "
                                   f"up to 256 with GroupNorm(8) (got {enc.encoder[0].out_channels})")
            x = _HipStem.apply(enc, x, *_hip_stem_params(enc))
            if x.shape[-2:] != (ho, wo):
                x = x.float()
        elif enc.use_encoder:
            x = x.float().contiguous(memory_format=torch.channels_last)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(amp)):
                x = torch.cat([enc._branch_train(x, enc.encoder), enc._branch_train(x, enc.sem_encoder)], dim=1)
            x = x.float()
        if x.shape[-2:] != (ho, wo):
            x = F.adaptive_avg_pool2d(x, output_size=(ho, wo))                 # naf.py:34
        if amp == "hip" and heads_rope == heads and (x.shape[1] // heads_rope) % 32 == 0:
            # RoPE, key pooling and their backward as HIP kernels too (naf_rope_pool_fwd / naf_rope_pool_bwd): the guidance stays
            # bf16 channels-last from the stem's last layer to the attention kernel, and so does its gradient on the way back
            if enc.rope.training and enc.rope.cache_train_coords:
                if (ho, wo) not in enc.rope._train_tables:
                    enc.rope._train_tables[(ho, wo)] = _rope_train_tables(enc.rope, ho, wo)
                tab_y, tab_x = enc.rope._train_tables[(ho, wo)]
            else:
                tab_y, tab_x = _rope_train_tables(enc.rope, ho, wo) if enc.rope.training else enc.rope.tables(ho, wo)
            xcl = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            q5, k5 = ops.RopePoolFunction.apply(xcl, tab_y.contiguous(), tab_x.contiguous(), heads_rope, (h, w))
            B, C = features.shape[:2]
            v5 = features.reshape(B, heads, C // heads, h, w).permute(0, 1, 3, 4, 2).to(torch.bfloat16).contiguous()
            out_dtype = torch.bfloat16 if features.dtype == torch.bfloat16 else torch.float32
            res = ops.XnaFunction.apply(q5, k5, v5, self.upsampler.kernel_size, self.upsampler.scale, out_dtype, bool(return_weights))
            out5, logits = res if return_weights else (res, None)
            # [B, heads, Ho, Wo, Dv] is a view of a [B, Ho, Wo, heads * Dv] buffer: hand it out as a logical NCHW view of that
            # (channels-last memory, no transpose copy of the largest tensor of the step)
            out = out5.permute(0, 2, 3, 1, 4).reshape(B, ho, wo, C).permute(0, 3, 1, 2)
            return (out, logits) if return_weights else out
        # RoPE (rope.py:15-34,139-153) from the cached tables: angle index t < D/4 -> row, else column
        # [Ho, 2, P], [Wo, 2, P]; in training mode with the reference's coordinate augmentation (rope.py:107-124)
        if enc.rope.training and enc.rope.cache_train_coords:
            if (ho, wo) not in enc.rope._train_tables:
                enc.rope._train_tables[(ho, wo)] = _rope_train_tables(enc.rope, ho, wo)
            tab_y, tab_x = enc.rope._train_tables[(ho, wo)]
        else:
            tab_y, tab_x = _rope_train_tables(enc.rope, ho, wo) if enc.rope.training else enc.rope.tables(ho, wo)
        B, Cq = x.shape[:2]
        D = Cq // heads_rope
        cos = torch.cat([tab_y[:, 0, None, :].expand(ho, wo, -1), tab_x[None, :, 0, :].expand(ho, wo, -1)], dim=-1)
        sin = torch.cat([tab_y[:, 1, None, :].expand(ho, wo, -1), tab_x[None, :, 1, :].expand(ho, wo, -1)], dim=-1)
        xh = x.reshape(B, heads_rope, D, ho, wo).permute(0, 1, 3, 4, 2)        # [B, n, Ho, Wo, D]
        x1, x2 = xh[..., : D // 2], xh[..., D // 2:]
        xr = torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)
        xr = xr.permute(0, 1, 4, 2, 3).reshape(B, Cq, ho, wo)
        if ho % h == 0 and wo % w == 0:                                        # naf.py:63-69 (pooled AFTER RoPE); box mean as a
            k = xr.reshape(B, Cq, h, ho // h, w, wo // w).mean(dim=(3, 5))    # reshape: its backward is a broadcast, not atomics
        else:
            k = F.adaptive_avg_pool2d(xr, output_size=(h, w))
        Dq, C = Cq // heads, features.shape[1]
        to5 = lambda t, d: t.reshape(B, heads, d, *t.shape[-2:]).permute(0, 1, 3, 4, 2).to(torch.bfloat16).contiguous()
        q5, k5, v5 = to5(xr, Dq), to5(k, Dq), to5(features, C // heads)
        out_dtype = torch.bfloat16 if features.dtype == torch.bfloat16 else torch.float32
        res = ops.XnaFunction.apply(q5, k5, v5, self.upsampler.kernel_size, self.upsampler.scale, out_dtype, bool(return_weights))
        out5, logits = res if return_weights else (res, None)
        out = out5.permute(0, 1, 4, 2, 3).reshape(B, C, ho, wo)
        return (out, logits) if return_weights else out

    def forward(self, image, features, output_size, return_weights=False, *args, **kwargs):
        """``naf(image, lr_features, target_size)`` (naf.py:104-116).  The reference's forward is always differentiable;
        here the fused inference kernels run unless a gradient is actually wanted: autograd enabled AND (an input requires
        grad, or the module is in ``.train()`` mode with trainable parameters) -- then the call is ``forward_train``
        (train.py:127-137, denoising.py:213 work unchanged).  README usage (``naf.eval()`` then ``naf(...)``) and any call
        under ``torch.no_grad()`` take the inference path; that path always uses the deterministic eval-mode RoPE
        coordinates (the reference's train-mode coordinate jitter only exists on the differentiable path here)."""
        if torch.is_grad_enabled() and (image.requires_grad or features.requires_grad or
                                        (self.training and any(p.requires_grad for p in self.parameters()))):
            # the library's own differentiable stem whenever it serves the width, with or without torch.autocast (round 6: the
            # bf16-activation contract is what the inference path computes); else the torch stem in the ambient precision.
            # return_weights (attentions.py:64-67 under autograd): (out, scores) with the scores as a non-differentiable output
            return self.forward_train(image, features, output_size, amp="auto", return_weights=return_weights)
        with torch.no_grad():
            return self._forward_inference(image, features, output_size, return_weights)

    def _forward_inference(self, image, features, output_size, return_weights=False):
        if not (image.is_cuda and features.is_cuda):
            raise RuntimeError("naf_amd.NAF runs only on a ROCm device (HIP kernels, no CPU fallback); got "
                               f"image on {image.device}, features on {features.device}")
        if image.dim() != 4 or features.dim() != 4 or image.shape[0] != features.shape[0]:
            raise ValueError(f"expected image [B,3,H,W] and features [B,C,h,w], got {tuple(image.shape)} / {tuple(features.shape)}")
        if image.shape[0] == 0:
            # an empty batch flows through the reference's torch ops as empty tensors; there is nothing to launch here
            ho, wo = int(output_size[0]), int(output_size[1])
            odt = features.dtype if features.dtype in (torch.bfloat16, torch.float32) else torch.float32
            out = torch.empty((0, ho, wo, features.shape[1]), dtype=odt, device=features.device).permute(0, 3, 1, 2)
            if out.dtype != features.dtype:
                out = out.to(features.dtype)
            if return_weights:
                k = self.upsampler.kernel_size
                return out, torch.empty((0, self.upsampler.num_heads, ho, wo, k[0] * k[1]), dtype=torch.float32, device=features.device)
            return out
        if self.single_call:
            plan = self._forward_plan(image, features, output_size)
            if plan is not None:
                timer = ops.KERNEL_TIMER
                ev = pe = None
                if timer is not None and getattr(timer, "enabled", False):
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    for e in ev:
                        e.record()          # creates the underlying hipEvent_t; naf_forward re-records it around the kernel
                    timer.pairs.setdefault("xna_mfma", []).append(ev)
                    every = int(getattr(timer, "phases", 1) or 0)       # 0 / False: none; n: every n-th call (an event record
                    timer._calls = getattr(timer, "_calls", -1) + 1       # between two kernels costs ~5 us of device time)
                    if every > 0 and timer._calls % every == 0:
                        # the phases of the ONE call (naf_forward_args.phase_events, version >= 105: the branches' layers alternate):
                        # the whole stem, both first convolutions, ONE launch of each block-layer kernel (stage 1), the RoPE /
                        # key-pooling pre-pass, the attention
                        pe = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
                        for e in pe:
                            e.record()
                        # under the A/B knob NAF_STEM_ORDER=0 (one branch after the other) the library never records [7] and [1] / [2] / [3]
                        # mean other things (include/naf_hip.h): only the boundaries both orders share are registered then
                        knobs = os.environ.get("NAF_HIP_KNOBS") == "1"
                        sequential = knobs and os.environ.get("NAF_STEM_ORDER") == "0"
                        one_stream = plan.planned_streams() == 1     # the library's plan for these shapes (naf_forward_streams)
                        pairs = (("stem", 0, 4), ("rope_pool", 4, 5), ("attention", 5, 6))
                        if sequential:
                            pass
                        elif one_stream:      # round 3: the branches' layers alternate on one stream
                            pairs += (("stem_first_convs", 0, 1), ("stem_layer_1x1", 2, 3), ("stem_layer_3x3", 3, 7))
                        else:                 # two streams (the host lends the second: ops.forward_aux); [2] .. [7] bracket one 3x3 launch (the 1x1 launches run beside it)
                            pairs += (("stem_first_convs", 0, 1), ("stem_layer_3x3", 2, 7))
                        for name, i, j in pairs:
                            timer.pairs.setdefault(name, []).append((pe[i], pe[j]))
                return plan.run(image, features, ev, return_logits=bool(return_weights), phase_events=pe)
        fuse_for = None
        if features.shape[1] % self.upsampler.num_heads == 0:
            fuse_for = (features.shape[1] // self.upsampler.num_heads,
                        torch.bfloat16 if features.dtype == torch.bfloat16 else torch.float32)
        q5, k5, tabs = self.guidance_qk(image, features.shape[-2:], output_size, fuse_for=fuse_for)
        with ops._Timed("attention"):
            return self.upsampler(q5, k5, features, return_weights=return_weights, path=self.xna_path, rope_tables=tabs)
