"""The reference's own backward speed sweep (test/backward_speed.py with test/test_utils.py: low-res grid 32^2, upsampling ratio in
{2, 4, 8, 16, 32}, embedding 384 / 1024, window 9): attention forward and backward kernels per ratio, which kernel serves each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
lr, ks, heads = 32, 9, 4
for C in (384, 1024):
    for ratio in (2, 4, 8, 16, 32):
        out = lr * ratio
        q = torch.randn(1, out, out, heads, 64, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        k = torch.randn(1, lr, lr, heads, 64, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        v = torch.randn(1, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        g = torch.randn(1, out, out, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        tf = timed(lambda: ops.xna_forward(q, k, v, ks))
        tb = timed(lambda: ops.xna_backward(q, k, v, g, ks))
        print("C %4d  32^2 -> %4d^2 (ratio %2d): forward %-7s %8.4f ms | backward %-7s %8.4f ms" % (
            C, out, ratio, ops.xna_select(q, k, v, ks), tf, ops.xna_backward_select(q, k, v, ks), tb))
