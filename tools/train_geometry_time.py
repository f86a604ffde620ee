"""The reference's OWN training step geometry (train.py:113-133 with config/base.yaml: img_size 512, ViT-B/16, down_factor fixed 0.5,
batch 4, window 9): guidance image 128^2, low-res features 768 x 16^2, target = the high-res feature grid 32^2 -- attention at
ratio 2.  Times the attention forward / backward kernels at that geometry and the whole step (forward_train + loss + backward + SGD)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import NAF, ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C, lr, out, ks, heads = 4, 768, 16, 32, 9, 4

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

q = torch.randn(B, out, out, heads, 64, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
k = torch.randn(B, lr, lr, heads, 64, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
v = torch.randn(B, lr, lr, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
g = torch.randn(B, out, out, heads, C // heads, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
print("attention %dx%d -> %dx%d, B %d, C %d, window %d: forward path %s %.4f ms | backward path %s %.4f ms" % (
    lr, lr, out, out, B, C, ks, ops.xna_select(q, k, v, ks), timed(lambda: ops.xna_forward(q, k, v, ks)),
    ops.xna_backward_select(q, k, v, ks), timed(lambda: ops.xna_backward(q, k, v, g, ks))))

m = NAF(kernel_size=ks).to(dev).train()
opt = torch.optim.SGD(m.parameters(), lr=1e-3)
img = torch.randn(B, 3, 128, 128, device=dev)
ft = torch.randn(B, C, lr, lr, device=dev)
tgt = torch.randn(B, C, out, out, device=dev)
for amp in (False, True, "hip"):
    def step():
        opt.zero_grad(set_to_none=True)
        o = m.forward_train(img, ft, (out, out), amp=amp)
        loss = (o.float() - tgt).pow(2).mean()
        loss.backward()
        opt.step()
        return loss
    print("whole step, amp=%s: %.3f ms" % (amp, timed(step, 10)))
if "--profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3): step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
