"""Time every stem kernel separately on a G1-shaped input (1 x 3 x 1024 x 1024): conv0 1x1 / 3x3 and the
GroupNorm+SiLU+conv layers 1x1 / 3x3.  python tools/stem_layer_bench.py [H W]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 1024)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B = 1
    img = torch.rand(B, 3, H, W, device=dev)
    y0 = torch.empty(B, H, W, 128, dtype=torch.bfloat16, device=dev)
    y1 = torch.empty_like(y0)
    st = ops.new_stats(B, dev, lead=(4,))
    gw, gb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    bias = torch.zeros(128, device=dev)
    for k in (1, 3):
        w0 = torch.randn(128, 3, k, k, device=dev) * 0.2
        b0 = torch.randn(128, device=dev) * 0.1
        t = timed(lambda: ops.stem_conv0(img, w0, b0, y0, st[0]))
        print("conv0 %dx%d            %.4f ms  (%.0f GB/s written)" % (k, k, t, B * H * W * 256 / t / 1e6))
    st.zero_(); ops.stem_conv0(img, w0, b0, y0, st[0])
    for k in (1, 3):
        wp = (torch.randn(k * k, 128, 128, device=dev) * (0.05 / k)).to(torch.bfloat16)
        t = timed(lambda: ops.stem_conv(y0, st[0], gw, gb, 1e-5, wp, bias, y1, st[1]))
        fl = 2.0 * B * H * W * 128 * 128 * k * k
        print("GN+SiLU+conv %dx%d     %.4f ms  (%.0f TFLOP/s, %.0f GB/s r+w)" % (k, k, t, fl / t / 1e9, B * H * W * 512 / t / 1e6))
main()
