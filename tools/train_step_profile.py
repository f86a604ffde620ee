import os, sys, torch
sys.path.insert(0, "/root/repo")
from naf_amd import NAF
import naf_amd.model as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = NAF().to(dev).train()
opt = torch.optim.SGD(m.parameters(), lr=1e-3)
img = torch.randn(1, 3, 448, 448, device=dev)
ft = torch.randn(1, 384, 28, 28, device=dev)
tgt = torch.randn(1, 384, 448, 448, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = m.forward_train(img, ft, (448, 448))
    loss = (out.float() - tgt).pow(2).mean()
    loss.backward()
    opt.step()
    return loss
def timeit(n=10):
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("fp32 stem: %.2f ms" % timeit())
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
