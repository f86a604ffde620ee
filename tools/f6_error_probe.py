"""Where does golden F6's (denoising-like: ratio 1, one head, C = 3, window 5) whole-forward error come from?
Prints the error distribution of the HIP forward against the reference's golden output, and the same with the attention fed by
the ORACLE's fp32 guidance (isolates the bf16-activation stem from the attention kernels).  Run on the GPU box."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import naf_oracle as O
from naf_amd import NAF, ops

g = np.load(os.path.join(ROOT, "tests", "golden", "F6_denoise_d1.npz"))
dev = torch.device("cuda:0")
dim, k = int(g["dim"]), int(g["k"])
p = O.make_params(dim=dim, heads_rope=1, seed=int(g["param_seed"]))
m = NAF(dim=dim, heads_attn=1, heads_rope=1, kernel_size=k).eval()
m.load_state_dict(p, strict=True)
m = m.to(dev)
shp = tuple(g["shape"])
img, ft = O.hash_normal(shp, int(g["image_seed"])), O.hash_normal(shp, int(g["feat_seed"]))
ref, ref_lg = torch.from_numpy(g["out"]), torch.from_numpy(g["logits"])
out, lg = m(img.to(dev), ft.to(dev), shp[-2:], return_weights=True)
out, lg = out.float().cpu(), lg.cpu()


def report(name, got, want):
    err = (got - want).abs()
    bound = 2e-2 + 1e-2 * want.abs()
    q = torch.quantile(err.flatten()[:4_000_000], torch.tensor([0.5, 0.9, 0.99, 0.999]))
    print(f"{name}: shape {tuple(got.shape)} |ref| max {float(want.abs().max()):.3f}  err mean {float(err.mean()):.3e} "
          f"p50 {q[0]:.2e} p90 {q[1]:.2e} p99 {q[2]:.2e} p99.9 {q[3]:.2e} max {float(err.max()):.3e}; "
          f"outside 2e-2+1e-2|ref|: {int((err > bound).sum())} of {err.numel()} ({100.0 * float((err > bound).float().mean()):.3f} %)")
    return err


print("dim", dim, "k", k, "shape", shp)
e_out = report("out     (HIP stem + HIP attention)", out, ref)
e_lg = report("logits  (HIP stem + HIP attention)", lg, ref_lg)
# softmax peakedness at the worst output elements
i = int(e_out.argmax())
b, c, y, x = np.unravel_index(i, e_out.shape)
pr = torch.softmax(ref_lg[b, 0, y, x], dim=-1)
pg = torch.softmax(lg[b, 0, y, x], dim=-1)
print(f"worst out element (b{b} c{c} y{y} x{x}): ref {float(ref[b, c, y, x]):.4f} got {float(out[b, c, y, x]):.4f}; softmax top-2 ref "
      f"{[round(float(v), 3) for v in pr.topk(2).values]} got {[round(float(v), 3) for v in pg.topk(2).values]}; "
      f"max |dlogit| there {float((lg[b, 0, y, x] - ref_lg[b, 0, y, x]).abs().max()):.3e}; values spread {float(ft[b, c].max() - ft[b, c].min()):.2f}")
# attention alone on the oracle's own (fp32) guidance: rope + pool by the oracle, bf16 q/k/v contract, HIP kernels
with torch.no_grad():
    xq = O.image_encoder(img, shp[-2:], p, 1)
    kk = O.key_pool(xq, shp[-2:])
    r2 = O.xna(xq.to(torch.bfloat16).float(), kk.to(torch.bfloat16).float(), ft.to(torch.bfloat16).float(), k, 1)
to5 = lambda t: t.view(t.shape[0], 1, t.shape[1], *t.shape[-2:]).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16).to(dev)
o2 = ops.xna_forward(to5(xq), to5(kk), to5(ft), k, out_dtype=torch.float32)
o2 = o2.permute(0, 1, 4, 2, 3).reshape(ref.shape).float().cpu()
report("out     (oracle guidance, bf16 q/k/v, HIP attention) vs golden", o2, ref)
report("out     (oracle guidance, bf16 q/k/v, HIP attention) vs oracle on the same bf16 q/k/v", o2, r2)
