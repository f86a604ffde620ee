"""Stress: many forwards over changing geometries / dtypes / batch sizes (plan caches, workspace reuse, table caches);
checks finiteness and that memory does not grow."""
import os, sys, torch, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd
dev = torch.device("cuda:0")
random.seed(0); torch.manual_seed(0)
models = {k: naf_amd.NAF(kernel_size=k).to(dev).eval() for k in (3, 7, 9, 15)}
base = None
for it in range(400):
    k = random.choice(list(models))
    h, w = random.randint(k, 24), random.randint(k, 24)
    mode = random.random()
    if mode < 0.4:
        d = random.choice([2, 4, 8, 14, 16, 16, 16, 32])
        H, W = h * d, w * d
    elif mode < 0.8:
        H, W = int(h * random.uniform(1.0, 12.0)), int(w * random.uniform(1.0, 12.0))
    else:
        H, W = h, w
    H, W = max(H, h, 2), max(W, w, 2)
    if k * (H // h) > H or k * (W // w) > W or H * W > 300 * 300:
        continue
    B = random.choice([1, 1, 2, 3])
    C = random.choice([64, 128, 384, 768, 24])
    ft = torch.randn(B, C, h, w, device=dev).to(random.choice([torch.bfloat16, torch.float32]))
    img = torch.randn(B, 3, H, W, device=dev)
    out = models[k](img, ft, (H, W))
    assert out.shape == (B, C, H, W) and torch.isfinite(out.float()).all(), (it, k, h, w, H, W, B, C)
    if it == 100:
        torch.cuda.synchronize(); base = torch.cuda.memory_allocated()
torch.cuda.synchronize()
print("ok: 400 iterations; allocated at it 100: %.1f MB, at the end: %.1f MB" % (base / 2**20, torch.cuda.memory_allocated() / 2**20))
