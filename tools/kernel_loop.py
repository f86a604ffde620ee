"""Loop ONE kernel (or the whole G1 forward) for a number of seconds so that rocm-smi can be sampled beside it (tools/power_sample.sh).
python tools/kernel_loop.py c1|c3|fwd|copy [seconds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops

def main():
    which = sys.argv[1]
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    H = W = 1024
    if which in ("c1", "c3"):
        k = 1 if which == "c1" else 3
        y0 = (torch.randn(1, H, W, 128, device=dev)).to(torch.bfloat16)
        y1 = torch.empty_like(y0)
        st = ops.new_stats(1, dev, lead=(2,))
        st[0, 0, :, :, 1] = H * W * 16.0
        gw, gb, bias = torch.ones(128, device=dev), torch.zeros(128, device=dev), torch.zeros(128, device=dev)
        wp = (torch.randn(k * k, 128, 128, device=dev) * (0.05 / k)).to(torch.bfloat16)
        fn = lambda: ops.stem_conv(y0, st[0], gw, gb, 1e-5, wp, bias, y1, st[1])
    elif which == "copy":
        a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
        fn = lambda: b.copy_(a)
    else:
        from naf_amd.model import NAF
        m = NAF().to(dev).eval()
        img = torch.rand(1, 3, H, W, device=dev)
        feats = torch.randn(1, 768, 64, 64, device=dev)
        with torch.no_grad():
            fn = lambda: m(img, feats, (H, W))
            fn()
    n = 0
    torch.cuda.synchronize()
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        while time.time() - t0 < secs:
            e0.record()
            for _ in range(200): fn()
            e1.record(); torch.cuda.synchronize()
            n += 1
            last = e0.elapsed_time(e1) / 200
    print("%s: %.4f ms per call (last block of 200), %d blocks" % (which, last, n))
main()
