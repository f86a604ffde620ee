"""The reference's denoising TRAINING step (denoising.py:209-220: model(noisy_norm, noisy, (S, S)) in train mode, loss, backward, optimizer) at its
model widths NAF(dim 96 ... 512, one head, window 15), batch 2 x 3 x 256 x 256: the three stem arms of forward_train -- "auto" (round 6: the library's
own differentiable stem at every width), amp=False (fp32 torch / MIOpen stem, what these widths ran until round 5), amp=True (autocast torch stem) --
and, with --profile, where the time of the "auto" step goes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd

dev = torch.device("cuda:0")
S, B = 256, 2


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for dim in (96, 256, 512):
    torch.manual_seed(0)
    m = naf_amd.NAF(dim=dim, heads_attn=1, heads_rope=1, kernel_size=15).to(dev).train()
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    noisy = torch.randn(B, 3, S, S, device=dev)
    clean = torch.randn(B, 3, S, S, device=dev)
    line = "NAF(dim %3d, 1 head, window 15), %d x 3 x %d^2:" % (dim, B, S)
    for amp in ("auto", False, True):
        def step():
            opt.zero_grad(set_to_none=True)
            out = m.forward_train(noisy, noisy, (S, S), amp=amp)
            loss = (out.float() - clean).pow(2).mean()
            loss.backward()
            opt.step()
        torch.cuda.reset_peak_memory_stats()
        t = timed(step)
        line += "   amp=%-5s %7.3f ms (%5.0f MB)" % (amp, t, torch.cuda.max_memory_allocated() / 2**20)
    print(line, flush=True)
    if "--profile" in sys.argv:
        from torch.profiler import profile, ProfilerActivity
        def step():
            opt.zero_grad(set_to_none=True)
            out = m.forward_train(noisy, noisy, (S, S), amp="auto")
            loss = (out.float() - clean).pow(2).mean()
            loss.backward()
            opt.step()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3): step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=90))
