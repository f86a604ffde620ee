"""Per-kernel time of the composed forward (ops.KERNEL_TIMER) at a few image sizes: where does a non-headline size
lose against G1's per-pixel rate?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd
from naf_amd import ops
from bench import EventTimer

dev = torch.device("cuda:0")
for (h, w, Ho, Wo, C, ks) in [(64, 64, 1024, 1024, 768, 7), (37, 37, 518, 518, 768, 9), (32, 32, 512, 512, 768, 9), (28, 28, 448, 448, 384, 9)]:
    m = naf_amd.NAF(kernel_size=ks).to(dev).eval()
    img = torch.randn(1, 3, Ho, Wo, device=dev)
    ft = torch.randn(1, C, h, w, device=dev)
    for single in (True, False):
        m.single_call = single
        for _ in range(3): m(img, ft, (Ho, Wo))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m(img, ft, (Ho, Wo))
        e1.record(); torch.cuda.synchronize()
        tot = e0.elapsed_time(e1) / 20
        if single:
            print("%dx%d -> %dx%d C%d k%d: single call %.3f ms (%.1f Mpix/s)" % (h, w, Ho, Wo, C, ks, tot, Ho * Wo / tot / 1e3))
    t = EventTimer(); t.enabled = True
    ops.KERNEL_TIMER = t
    for _ in range(10): m(img, ft, (Ho, Wo))
    torch.cuda.synchronize()
    ops.KERNEL_TIMER = None
    parts = {k: (t.mean_ms(k), len(v) // 10) for k, v in t.pairs.items()}
    print("   composed %.3f ms; kernels: " % tot + "  ".join("%s %.3f x%d" % (k, v[0], v[1]) for k, v in parts.items()) +
          "  | sum %.3f" % sum(v[0] * v[1] for v in parts.values()))
