"""Per-layer error budget of the bf16-activation HIP conv stem against the fp32 oracle ("quantify per-layer
error growth and say why 2e-2 is unreachable, or reach it").  Run on the GPU box:  python tools/stem_error_budget.py [size]

For both branches and every layer l (conv0, then four GroupNorm -> SiLU -> conv layers):
  total : HIP chain output after layer l  vs  the fp32 oracle chain (exact inputs all the way)
  layer : HIP layer l output              vs  the fp32 layer applied to the HIP chain's OWN layer l-1 output
          (isolates what ONE layer adds: bf16 weights, bf16 rounding of the activated input, bf16 rounding of the output)
Errors as mean / max |err| and relative to the layer's output RMS."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import NAF, ops          # noqa: E402
from oracle import naf_oracle as O    # noqa: E402  (tools/ may use the checker; the product never does)


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    p = O.make_params(seed=9)
    m = NAF().eval()
    m.load_state_dict(p, strict=True)
    m = m.to(dev)
    enc = m.image_encoder
    img = O.hash_normal((1, 3, S, S), 91)
    imgd = img.to(dev)
    print(f"image 1x3x{S}x{S}, default NAF (hidden 128), weights O.make_params(seed=9)")
    print(f"{'branch':8s} {'layer':6s} | {'total mean':>10s} {'total max':>10s} {'rel rms':>8s} | {'layer mean':>10s} {'layer max':>10s} {'rel rms':>8s} | {'out rms':>8s}")
    for br, (seq, pre) in enumerate(((enc.encoder, "image_encoder.encoder"), (enc.sem_encoder, "image_encoder.sem_encoder"))):
        ks = seq[0].kernel_size[0]
        stats = ops.new_stats(1, dev, lead=(5,))
        bufs = [torch.empty((1, S, S, 128), dtype=torch.bfloat16, device=dev) for _ in range(2)]
        w0, b0 = seq[0].weight.detach().float().contiguous(), seq[0].bias.detach().float()
        ops.stem_conv0(imgd, w0, b0, bufs[0], stats[0])
        cur = bufs[0]
        # exact chain
        ex = O._conv_reflect(img, p[f"{pre}.0.weight"], p[f"{pre}.0.bias"])
        got = cur.permute(0, 3, 1, 2).float().cpu()
        rms = float(ex.pow(2).mean().sqrt())
        e = (got - ex).abs()
        print(f"{'1x1' if ks == 1 else '3x3':8s} {'conv0':6s} | {float(e.mean()):10.3e} {float(e.max()):10.3e} {float(e.pow(2).mean().sqrt()) / rms:8.1e} | "
              f"{float(e.mean()):10.3e} {float(e.max()):10.3e} {float(e.pow(2).mean().sqrt()) / rms:8.1e} | {rms:8.3f}")
        st = 0
        for bi, blk in enumerate(list(seq)[1:], start=1):
            for ni, (norm, conv) in enumerate(((blk.norm1, blk.conv1), (blk.norm2, blk.conv2)), start=1):
                st += 1
                prev = cur.permute(0, 3, 1, 2).float().cpu()                    # the HIP chain's own input to this layer
                dst = bufs[st % 2]
                ops.stem_conv(cur, stats[st - 1], norm.weight.detach().float(), norm.bias.detach().float(), norm.eps, enc._packed(conv),
                              conv.bias.detach().float(), dst, stats[st] if st < 4 else None)
                cur = dst
                gw, gb = p[f"{pre}.{bi}.norm{ni}.weight"], p[f"{pre}.{bi}.norm{ni}.bias"]
                w, b = p[f"{pre}.{bi}.conv{ni}.weight"], p[f"{pre}.{bi}.conv{ni}.bias"]
                f = lambda t: O._conv_reflect(F.silu(F.group_norm(t, 8, gw, gb, eps=1e-5)), w, b)
                ex = f(ex)
                one = f(prev)
                got = cur.permute(0, 3, 1, 2).float().cpu()
                rms = float(ex.pow(2).mean().sqrt())
                et, el = (got - ex).abs(), (got - one).abs()
                print(f"{'':8s} {'b%dc%d' % (bi, ni):6s} | {float(et.mean()):10.3e} {float(et.max()):10.3e} {float(et.pow(2).mean().sqrt()) / rms:8.1e} | "
                      f"{float(el.mean()):10.3e} {float(el.max()):10.3e} {float(el.pow(2).mean().sqrt()) / rms:8.1e} | {rms:8.3f}")
    # what the stem error does to the output: whole forward vs oracle on the P1-like geometry at this size
    ft = O.hash_normal((1, 384, S // 16, S // 16), 92)
    out = m(imgd, ft.to(dev), (S, S)).float().cpu()
    ref = O.naf_forward_fast(p, img, ft, (S, S), kernel_size=9)
    e = (out - ref).abs()
    q = torch.quantile(e.flatten()[:: max(1, e.numel() // 4_000_000)], torch.tensor([0.5, 0.99, 0.9999]))
    print(f"whole forward (384 x {S // 16}^2 features -> {S}^2, window 9, fp32 out): mean |err| {float(e.mean()):.3e}, median {float(q[0]):.3e}, "
          f"p99 {float(q[1]):.3e}, p99.99 {float(q[2]):.3e}, max {float(e.max()):.3e}; fraction over 2e-2 + 1e-2|ref|: "
          f"{float((e > 2e-2 + 1e-2 * ref.abs()).float().mean()):.2e}; over 6e-2 + 3e-2|ref|: {float((e > 6e-2 + 3e-2 * ref.abs()).float().mean()):.2e}")
    # the same forward with the fp32 torch stem (no bf16 activations): isolates the attention's own bf16 contract
    x = O.rope(O.conv_stem(img, p), p["image_encoder.rope.periods"], 4)
    k = O.key_pool(x, (S // 16, S // 16))
    bf = lambda t: t.to(torch.bfloat16).float()
    ref_bf = O.xna_lowres(bf(x), bf(k), bf(ft), 9, 4)
    e2 = (ref_bf - ref).abs()
    print(f"oracle with only q / k / v rounded to bf16 (fp32 stem):                     mean |err| {float(e2.mean()):.3e}, max {float(e2.max()):.3e}; "
          f"fraction over 2e-2 + 1e-2|ref|: {float((e2 > 2e-2 + 1e-2 * ref.abs()).float().mean()):.2e}")


if __name__ == "__main__":
    main()
