"""The reference's denoising configuration end to end (denoising.py:213: NAF(dim, heads 1, window 15) on a 3-channel image
that is also the value tensor): forward time and where it goes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
for dim in (96, 256):
    m = naf_amd.NAF(dim=dim, heads_attn=1, heads_rope=1, kernel_size=15).to(dev).eval()
    x = torch.randn(2, 3, 256, 256, device=dev)
    for _ in range(3): m(x, x, (256, 256))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): m(x, x, (256, 256))
    e1.record(); torch.cuda.synchronize()
    print("dim %d: forward %.3f ms (batch 2 x 256^2)" % (dim, e0.elapsed_time(e1) / 10))
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3): m(x, x, (256, 256))
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=70))
