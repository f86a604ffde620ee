"""One training step (forward_train + backward + SGD) at the reference's published point (BASELINE.md:
image 448^2, 384 x 28^2 features -> 448^2, window 9: fwd+bwd+SGD 163.08 ms, 6016.5 MB on an A100-40GB).
The attention forward/backward are the HIP kernels; the conv stem, RoPE and pooling are torch ops (autograd)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import NAF

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = NAF().to(dev).train()
opt = torch.optim.SGD(m.parameters(), lr=1e-3)
img = torch.randn(1, 3, 448, 448, device=dev)
ft = torch.randn(1, 384, 28, 28, device=dev)
tgt = torch.randn(1, 384, 448, 448, device=dev)

AMP = "hip" if "--hip" in sys.argv else ("--amp" in sys.argv)   # --hip: the library's own differentiable stem (_HipStem)

def step():
    opt.zero_grad(set_to_none=True)
    out = m.forward_train(img, ft, (448, 448), amp=AMP)
    loss = (out.float() - tgt).pow(2).mean()
    loss.backward()
    opt.step()
    return loss

for _ in range(3): step()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n): l = step()
e1.record(); torch.cuda.synchronize()
print(("HIP stem (bf16 activations) " if AMP == "hip" else "amp (bf16 stem convs) " if AMP else "") + "fwd+bwd+SGD step: %.2f ms   peak memory %.0f MB   (reference, A100-40GB: 163.08 ms, 6016.5 MB)   loss %.4f"
      % (e0.elapsed_time(e1) / n, torch.cuda.max_memory_allocated() / 2**20, float(l)))

if "--profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3): step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
