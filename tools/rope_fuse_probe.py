"""A/B of rotate-on-load in naf_xna_fwd on G1 shapes: query layout (head-major / channels-last) x rotation
(materialised by naf_rope_pool_fwd / applied on load)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = torch.device("cuda:0")
B, heads, Dq, C, lr, out, ks = 1, 4, 64, 768, 64, 1024, 7
torch.manual_seed(0)
x = torch.randn(B, out, out, heads * Dq, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)      # channels-last
per = 100.0 ** (2 * torch.arange(16, dtype=torch.float32, device=dev) / 32)
ty, tx = ops.rope_tables(per, out, out)
q_hm, k5 = ops.rope_pool(x, ty, tx, heads, (lr, lr), q_layout="head_major")
q_cl, _ = ops.rope_pool(x, ty, tx, heads, (lr, lr), q_layout="channels_last")
raw_cl = x.permute(0, 2, 3, 1).unflatten(3, (heads, Dq)).permute(0, 3, 1, 2, 4)
raw_hm = raw_cl.contiguous()
v5 = torch.randn(B, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
o = torch.empty((B, out, out, heads, C // heads), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
cfgs = (("materialised, head-major", q_hm, None), ("materialised, channels-last", q_cl, None),
        ("rotate-on-load, head-major", raw_hm, (ty, tx)), ("rotate-on-load, channels-last", raw_cl, (ty, tx)))
res = {n: [] for n, _, _ in cfgs}
for rnd in range(6):
    for name, q, tabs in cfgs:
        res[name].append(timed(lambda: ops.xna_forward(q, k5, v5, ks, out=o, path="mfma", rope_tables=tabs), n=10))
for name, ts in res.items():
    ts = sorted(ts)
    print("%-32s min %.4f  median %.4f  max %.4f ms" % (name, ts[0], ts[len(ts) // 2], ts[-1]))
