import sys, torch
sys.path.insert(0, "/root/repo")
from naf_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
heads, out, lr = 4, 1024, 64
q = torch.randn(1, heads, out, out, 64, device=dev).to(torch.bfloat16)
k = torch.randn(1, heads, lr, lr, 64, device=dev).to(torch.bfloat16)
for C in (384, 768, 1024):
  v = torch.randn(1, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
  for ks in (7, 9):
    for dt in (torch.bfloat16, torch.float32):
        o = torch.empty((1, out, out, heads, C // heads), dtype=dt, device=dev).permute(0, 3, 1, 2, 4)
        t = timed(lambda: ops.xna_forward(q, k, v, ks, out=o, out_dtype=dt, path="mfma"))
        nb = out * out * (256 * 2 + C * (2 if dt == torch.bfloat16 else 4))
        print("C %4d k %d %-14s %.4f ms  %.0f GB/s" % (C, ks, str(dt), t, nb / t / 1e6))
