"""Time naf_xna_fwd / naf_xna_bwd on BASELINE shapes (attention only).  python tools/xna_bwd_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops

def timed(fn, n=int(os.environ.get("BWD_BENCH_N", "10"))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = torch.device("cuda:0")
heads, Dq = 4, 64
for name, C, lr, out, ks in (("G1", 768, 64, 1024, 7), ("G2-k7", 1024, 32, 512, 7), ("G2-k11", 1024, 32, 512, 11), ("G2-k15", 1024, 32, 512, 15), ("512^2 C1024 k13", 1024, 32, 512, 13), ("G3 (1 image)", 1024, 64, 1024, 7), ("k9 C384", 384, 64, 1024, 9), ("k9 C768", 768, 64, 1024, 9), ("448^2 C768 k9", 768, 28, 448, 9), ("k9 C1024", 1024, 64, 1024, 9), ("448^2 C1024 k9", 1024, 28, 448, 9), ("448^2 C384 k9", 384, 28, 448, 9)):
    q = torch.randn(1, heads, out, out, Dq, device=dev).to(torch.bfloat16)
    k = torch.randn(1, heads, lr, lr, Dq, device=dev).to(torch.bfloat16)
    v = torch.randn(1, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    g = torch.randn(1, out, out, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    tf = timed(lambda: ops.xna_forward(q, k, v, ks, path="mfma"))
    if not ops.xna_backward_supported(q, k, v, ks):
        print("%-16s fwd %.3f ms   bwd unsupported" % (name, tf)); continue
    tb = timed(lambda: ops.xna_backward(q, k, v, g, ks))
    # bytes the backward must move: read q, dout, write dq (+ low-res tensors)
    gb = out * out * (2 * 256 + C) * 2 / 1e9
    fl = 2.0 * ks * ks * out * out * (2 * 256 + 3 * C)   # S, dP, dQ, dK, dV contractions (unpadded, each once)
    print("%-16s fwd %.3f ms   bwd %.3f ms  (%.0f GB/s of q+dout+dq, %.0f TFLOP/s)" % (name, tf, tb, gb / tb * 1e3, fl / tb / 1e9))
