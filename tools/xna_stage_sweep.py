"""Staged (whole-row) vs unstaged stores of the MFMA attention kernel across window / Dv combinations.
Run under NAF_XNA_STAGE=0 and =1 (planner override) and compare."""
import os, sys, torch
os.environ.setdefault("NAF_HIP_KNOBS", "1")   # the A/B knobs below are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for out, lr in ((1024, 64), (512, 32)):
    q = torch.randn(1, heads, out, out, Dq, device=dev).to(torch.bfloat16)
    k = torch.randn(1, heads, lr, lr, Dq, device=dev).to(torch.bfloat16)
    for C in (384, 768, 1024):
        v = torch.randn(1, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
        o = torch.empty((1, out, out, heads, C // heads), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
        for ks in (3, 5, 7, 9, 11, 15):
            t = timed(lambda: ops.xna_forward(q, k, v, ks, out=o, path="mfma"))
            gb = (out * out * (256 + C) * 2 + lr * lr * (256 + C) * 2) / 1e9
            print("out %4d lr %3d C %4d k %2d  %.4f ms  %.0f GB/s" % (out, lr, C, ks, t, gb / t * 1e3))
