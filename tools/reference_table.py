"""Every forward-time / memory row the reference publishes (test/test_results.json, A100-40GB) re-measured
here with the reference's protocol (test/forward_speed.py:31-52, test/test_utils.py:78-82: B = 1, fp32 tensors, default
NAF() with window 9, 5 warm-ups, 10 calls, events + synchronize + empty_cache() around every call)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd

dev = torch.device("cuda:0")
ROWS = [  # (label, img_size, embed_dim, ratio, lr_size, published ms, published MB or None)
    ("448^2 C384 (x16)", 448, 384, 16, 28, 56.24, 1786.5),
    ("448^2 C128", 448, 128, 16, 28, 50.80, None),
    ("448^2 C768", 448, 768, 16, 28, 88.83, 2669.7),
    ("448^2 C1024", 448, 1024, 16, 28, 104.49, 3258.4),
    ("out 56^2 from image 448^2", 448, 384, 2, 28, 39.51, None),
    ("out 112^2 from image 448^2", 448, 384, 4, 28, 40.17, None),
    ("out 224^2 from image 448^2", 448, 384, 8, 28, 42.51, None),
    ("896^2 (x32)", 896, 384, 32, 28, 267.94, 7101.5),
]
model = naf_amd.NAF().to(dev).eval()          # defaults: dim 256, 4 heads, window 9
print("%-28s %12s %12s %8s %12s %12s" % ("row", "published ms", "here ms", "ratio", "publ. MB", "here MB"))
for label, img_size, C, ratio, lr, pub_ms, pub_mb in ROWS:
    img = torch.randn(1, 3, img_size, img_size, device=dev)
    ft = torch.randn(1, C, lr, lr, device=dev)
    out_size = (ratio * lr, ratio * lr)
    for _ in range(5):
        with torch.no_grad():
            torch.cuda.empty_cache()
            _ = model(img, ft, out_size)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    total = 0.0
    per_call = []
    for _ in range(10):
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        s.record()
        with torch.no_grad():
            _ = model(img, ft, out_size)
        e.record(); torch.cuda.synchronize()
        total += s.elapsed_time(e)
        per_call.append(s.elapsed_time(e))
    # peak memory the way test/forward_memory-style harnesses read it: reset, one call, max allocated
    del _
    torch.cuda.empty_cache(); torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        o = model(img, ft, out_size)
    torch.cuda.synchronize()
    mb = torch.cuda.max_memory_allocated() / 2**20
    del o
    ms = total / 10
    print("%-28s %12.2f %12.3f %7.0fx %12s %12.1f" % (label, pub_ms, ms, pub_ms / ms, "%.1f" % pub_mb if pub_mb else "-", mb))
    if os.environ.get("NAF_TABLE_DEBUG"): print("      per call:", " ".join("%.2f" % t for t in per_call))
