import os, sys, torch
sys.path.insert(0, os.getcwd())
from naf_amd import ops
def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
dev = torch.device("cuda:0"); heads, Dq = 4, 64
for name, C, lr, out, ks in (("G2-k7", 1024, 32, 512, 7), ("G2-k9", 1024, 32, 512, 9), ("G2-k11", 1024, 32, 512, 11), ("G2-k13 C512", 512, 32, 512, 13), ("G2-k15", 1024, 32, 512, 15), ("k11 C384 1024^2", 384, 64, 1024, 11)):
    q = torch.randn(1, heads, out, out, Dq, device=dev).to(torch.bfloat16)
    k = torch.randn(1, heads, lr, lr, Dq, device=dev).to(torch.bfloat16)
    v = torch.randn(1, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    g = torch.randn(1, out, out, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    tf = timed(lambda: ops.xna_forward(q, k, v, ks))
    tb = timed(lambda: ops.xna_backward(q, k, v, g, ks))
    print("%-18s fwd %.3f ms   bwd %.3f ms (%s)" % (name, tf, tb, ops.xna_backward_select(q, k, v, ks)))
