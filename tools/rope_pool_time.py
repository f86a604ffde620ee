import sys, torch
sys.path.insert(0, "/root/repo")
from naf_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(1, 1024, 1024, 256, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
periods = 100.0 ** (2 * torch.arange(16, device=dev, dtype=torch.float32) / 32)
ty, tx = ops.rope_tables(periods, 1024, 1024)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("keys only %.4f ms   with q %.4f ms" % (t(lambda: ops.rope_pool(x, ty, tx, 4, (64, 64), write_q=False)), t(lambda: ops.rope_pool(x, ty, tx, 4, (64, 64)))))
