"""The reference's denoising call shape (denoising.py:213: ratio 1, C = 3, one head, window 15, dim 96-512): attention
kernel time per path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for S, Dq, ks in ((128, 96, 15), (256, 96, 15), (256, 256, 15), (256, 512, 15), (256, 256, 7)):
    q = torch.randn(1, 1, S, S, Dq, device=dev).to(torch.bfloat16)
    k = torch.randn(1, 1, S, S, Dq, device=dev).to(torch.bfloat16)
    v = torch.randn(1, S, S, 1, 3, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    path = ops.xna_select(q, k, v, ks)
    t = timed(lambda: ops.xna_forward(q, k, v, ks, out_dtype=torch.float32))
    fl = 2.0 * S * S * ks * ks * (Dq + 3)
    print("S %4d Dq %3d k %2d  path %-8s %8.3f ms  %.2f TFLOP/s" % (S, Dq, ks, path, t, fl / t / 1e9))
    g = torch.randn(1, S, S, 1, 3, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    bsel = ops.xna_backward_select(q, k, v, ks)
    tb = timed(lambda: ops.xna_backward(q, k, v, g, ks))
    line = "        backward: %-8s %8.3f ms" % (bsel, tb)
    if bsel == "rows" and S <= 128:
        line += "   (table-driven scalar kernel: %8.3f ms)" % timed(lambda: ops.xna_backward(q, k, v, g, ks, path="generic"), n=1)
    print(line)
