"""Attention kernel and whole forward across the geometries callers of the reference use besides the headline
ones: patch-14 / patch-8 ratios, small ratios, non-integer ratios, ratio 1.
Prints which attention path served each case, its time and algorithmic GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd
from naf_amd import ops

def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = torch.device("cuda:0")
heads, Dq = 4, 64
CASES = [  # (lr_h, lr_w, out_h, out_w, C, k)
    (64, 64, 1024, 1024, 768, 7),     # G1, d=16
    (64, 64, 512, 512, 768, 7),       # d=8
    (37, 37, 518, 518, 768, 9),       # patch 14
    (37, 37, 518, 518, 384, 9),
    (64, 64, 256, 256, 768, 7),       # d=4
    (128, 128, 256, 256, 768, 7),     # d=2
    (32, 48, 512, 768, 1024, 7),      # non-square, d=16
    (28, 28, 448, 448, 384, 9),       # REF448
    (28, 28, 128, 128, 384, 9),       # non-integer 4.57
    (28, 28, 64, 64, 384, 9),         # non-integer 2.29
    (37, 37, 512, 512, 768, 9),       # non-integer 13.8
    (256, 256, 256, 256, 384, 7),     # ratio 1
    (256, 256, 256, 256, 384, 15),
]
print("%-28s %-8s %9s %9s | %9s %9s" % ("case", "path", "attn ms", "GB/s", "fwd ms", "Mpix/s"))
for (h, w, Ho, Wo, C, ks) in CASES:
    q = torch.randn(1, heads, Ho, Wo, Dq, device=dev).to(torch.bfloat16)
    k = torch.randn(1, heads, h, w, Dq, device=dev).to(torch.bfloat16)
    v = torch.randn(1, h, w, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    o = torch.empty((1, Ho, Wo, heads, C // heads), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
    path = ops.xna_select(q, k, v, ks)
    t = timed(lambda: ops.xna_forward(q, k, v, ks, out=o))
    alt = []
    for pth in ("mfma", "union", "generic"):
        if pth == path or (pth == "generic" and Ho * Wo > 300 * 300):
            continue
        try:
            alt.append("%s %.4f" % (pth, timed(lambda: ops.xna_forward(q, k, v, ks, out=o, path=pth), n=5)))
        except Exception:
            pass
    gb = (Ho * Wo * (256 + C) * 2 + h * w * (256 + C) * 2) / 1e9
    model = naf_amd.NAF(kernel_size=ks).to(dev).eval()
    img = torch.randn(1, 3, Ho, Wo, device=dev)
    feat = torch.randn(1, C, h, w, device=dev)
    with torch.no_grad():
        tf = timed(lambda: model(img, feat, (Ho, Wo)))
    print("%3dx%-3d->%4dx%-4d C%-4d k%-2d  %-8s %9.4f %9.0f | %9.3f %9.1f | %s" % (h, w, Ho, Wo, C, ks, path, t, gb / t * 1e3, tf, Ho * Wo / tf / 1e3, "  ".join(alt)))
