"""Attention forward at the reference's default window (9 x 9): the planner's choice against the forced alternatives
(NAF_HIP_KNOBS=1 NAF_XNA_STAGE=0: never the staged cell kernel -> the sliding kernel; NAF_XNA_SLIDE=0: never the sliding kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
dev = torch.device("cuda:0"); heads, Dq = 4, 64
for name, C, lr, out, ks in (("G2-k9 C1024", 1024, 32, 512, 9), ("k9 C1024 1024^2", 1024, 64, 1024, 9), ("k9 C768 1024^2", 768, 64, 1024, 9), ("k9 C384 1024^2", 384, 64, 1024, 9), ("k9 C384 448^2", 384, 28, 448, 9), ("G2-k7", 1024, 32, 512, 7), ("G1", 768, 64, 1024, 7)):
    q = torch.randn(1, heads, out, out, Dq, device=dev).to(torch.bfloat16)
    k = torch.randn(1, heads, lr, lr, Dq, device=dev).to(torch.bfloat16)
    v = torch.randn(1, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    print("%-18s fwd %.4f ms" % (name, timed(lambda: ops.xna_forward(q, k, v, ks))), flush=True)
