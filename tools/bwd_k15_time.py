"""Large windows at integer ratios (15 x 15 = BASELINE configs[2]'s largest window): round 3 routed them to the row-streaming matrix-core
backward instead of the scalar table-driven kernel; since the end of round 5 the cell kernel takes them in channel chunks (xna_bwd.hip).
Every kernel that serves a shape is timed (naf_xna_bwd_args.path)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for lr, out, C, ks in ((32, 512, 1024, 11), (64, 1024, 768, 11), (32, 512, 1024, 15), (32, 512, 256, 15), (32, 512, 1024, 13), (32, 512, 768, 13), (64, 1024, 768, 15), (32, 448, 384, 9), (16, 32, 768, 9)):
    heads = 4
    q = torch.randn(1, out, out, heads, 64, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    k = torch.randn(1, lr, lr, heads, 64, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    v = torch.randn(1, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    g = torch.randn(1, out, out, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    sel = ops.xna_backward_select(q, k, v, ks)
    t = timed(lambda: ops.xna_backward(q, k, v, g, ks))
    line = "%3d^2 -> %4d^2  C %4d  k %2d: backward path %-7s %9.3f ms" % (lr, out, C, ks, sel, t)
    if "--fast" in sys.argv:
        print(line); continue
    if sel == "mfma":
        line += "   (row-streaming kernel: %9.3f ms)" % timed(lambda: ops.xna_backward(q, k, v, g, ks, path="rows"))
    if out <= 512:
        line += "   (scalar table-driven kernel: %9.3f ms)" % timed(lambda: ops.xna_backward(q, k, v, g, ks, path="generic"), n=1)
    print(line)
