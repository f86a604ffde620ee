"""Host time of one naf(image, feats, size) call (Python + ctypes + the launches of naf_forward): a 64 x 64 forward is GPU-trivial, so
the wall time per call of a long loop is what the host spends.  python tools/host_overhead.py"""
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd
dev = torch.device("cuda:0")
naf = naf_amd.NAF().to(dev).eval()
for S in (160, 256):
    img = torch.randn(1, 3, S, S, device=dev); ft = torch.randn(1, 768, S // 16, S // 16, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        for _ in range(20): naf(img, ft, (S, S))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000): naf(img, ft, (S, S))
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%d^2: %.1f us per call enqueued, %.1f us per call including the drain" % (S, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): naf(img, ft, (S, S))
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
