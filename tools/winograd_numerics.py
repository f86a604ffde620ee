#!/usr/bin/env python3
"""Price the NUMERICS of a Winograd F(2x2,3x3) form of the stem's GroupNorm -> SiLU -> Conv3x3(128 -> 128)
layers on the CPU, under the HIP stem's own contract (bf16 activations
between layers, fp32 accumulation, GroupNorm statistics from the fp32 results).

Three stems of the 3x3 branch are evaluated on the same image and parameters:
  oracle   : fp32 everywhere (oracle/naf_oracle.py -- this script is a measurement tool, like bench.py's CPU leg)
  direct   : the shipped contract -- activated input and weights rounded to bf16, products accumulated in fp32
  winograd : U = G g G^T in fp32 rounded ONCE to bf16; V = B^T d B in fp32 on the activated bf16 tile, ONE rounding
             to bf16; 16 point-wise [oc x ic] x [ic x tiles] products accumulated in fp32; output A^T M A in fp32
Reported: per-layer error of one layer on identical bf16 input (vs F.conv2d fp32 on that input), and the whole
branch against the fp32 oracle (the quantity tests hold to mean <= 8e-3).
"""
import argparse
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import naf_oracle as O  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def conv_direct(a_bf, w, b):
    x = F.pad(a_bf, (1, 1, 1, 1), mode="reflect")
    return F.conv2d(x, bf(w), b)


def conv_winograd(a_bf, w, b, round_v=True, round_u=True):
    Bn, C, H, W = a_bf.shape
    assert H % 2 == 0 and W % 2 == 0
    x = F.pad(a_bf, (1, 1, 1, 1), mode="reflect")
    U = torch.einsum("ai,ocij,bj->abco", G, w, G)           # [4,4,ic,oc]
    if round_u:
        U = bf(U)
    t = x.unfold(2, 4, 2).unfold(3, 4, 2)                    # [B,C,th,tw,4,4]
    V = torch.einsum("ai,nchwij,bj->abnchw", BT, t, BT)      # [4,4,B,C,th,tw]
    if round_v:
        V = bf(V)
    M = torch.einsum("abco,abnchw->abnohw", U, V)           # fp32 accumulation over ic
    Y = torch.einsum("ia,abnohw,jb->nohiwj", AT, M, AT)      # [B,oc,th,2,tw,2]
    return Y.reshape(Bn, -1, H, W) + b[None, :, None, None]


def branch(image, p, pre, conv, first_exact=True):
    x = O._conv_reflect(image, p[f"{pre}.0.weight"], p[f"{pre}.0.bias"])
    per_layer = []
    blk = 1
    while f"{pre}.{blk}.conv1.weight" in p:
        for j in (1, 2):
            xin = bf(x) if conv is not None else x          # the layer's input as the next kernel reads it (bf16)
            # GroupNorm statistics come from the fp32 results of the producing layer
            mu = x.reshape(x.shape[0], 8, -1).mean(-1)
            var = x.reshape(x.shape[0], 8, -1).var(-1, unbiased=False)
            g = p[f"{pre}.{blk}.norm{j}.weight"]
            be = p[f"{pre}.{blk}.norm{j}.bias"]
            xn = (xin.reshape(x.shape[0], 8, -1) - mu[..., None]) * torch.rsqrt(var[..., None] + 1e-5)
            xn = xn.reshape(x.shape) * g[None, :, None, None] + be[None, :, None, None]
            a = F.silu(xn)
            w, b = p[f"{pre}.{blk}.conv{j}.weight"], p[f"{pre}.{blk}.conv{j}.bias"]
            if conv is None:
                x = O._conv_reflect(a, w, b)
            else:
                a = bf(a)
                x = conv(a, w, b)
                ref = O._conv_reflect(a, w, b)               # fp32 weights, the same bf16 input
                per_layer.append((float((x - ref).abs().mean()), float((x - ref).abs().max()), float(ref.pow(2).mean().sqrt())))
        blk += 1
    return x, per_layer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--seeds", type=int, default=3)
    args = ap.parse_args()
    torch.set_num_threads(8)
    pre = "image_encoder.sem_encoder"
    print(f"# Winograd F(2x2,3x3) numerics, 3x3 branch of the stem, {args.size}^2 image, hash_normal parameters")
    for seed in range(args.seeds):
        p = O.make_params(seed=seed)
        img = O.hash_normal((1, 3, args.size, args.size), 77 + seed)
        ref, _ = branch(img, p, pre, None)
        rms = float(ref.pow(2).mean().sqrt())
        for name, conv in (("direct  ", conv_direct), ("winograd", conv_winograd),
                           ("wino, V fp32", lambda a, w, b: conv_winograd(a, w, b, round_v=False)),
                           ("wino, U fp32", lambda a, w, b: conv_winograd(a, w, b, round_u=False))):
            out, per = branch(img, p, pre, conv)
            e = (bf(out) - ref).abs()
            pl = "  ".join("%.2e/%.2e" % (m, mx) for m, mx, _ in per)
            print(f"seed {seed} {name:13s} branch vs fp32 oracle: mean {float(e.mean()):.3e} max {float(e.max()):.3e} (rms {rms:.3f})"
                  f" | per layer vs fp32-weight conv on the same bf16 input, mean/max: {pl}")


if __name__ == "__main__":
    main()
