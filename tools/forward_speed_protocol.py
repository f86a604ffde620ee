"""The reference's own timing protocol (test/forward_speed.py:31-52: 5 warm-ups, 10 timed calls, device events around each
call, a synchronize and torch.cuda.empty_cache() before every call) applied to this implementation, next to the
back-to-back protocol of bench.py.  Like-for-like with BASELINE.md's 56.24 ms (image 448^2, 384 x 28^2 features, window 9,
fp32 tensors as in test/test_utils.py:78-82)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd

dev = torch.device("cuda:0")
NUM_RUNS = 10
for name, (C, lr, out, ks) in {"REF448 (reference's point)": (384, 28, 448, 9), "G1": (768, 64, 1024, 7)}.items():
    model = naf_amd.NAF(kernel_size=ks).to(dev).eval()
    img = torch.randn(1, 3, out, out, device=dev)
    ft = torch.randn(1, C, lr, lr, device=dev)          # fp32 features -> fp32 output, as the reference's harness
    for variant in ("reference protocol (empty_cache + sync per call)", "sync per call, no empty_cache", "back to back"):
        for _ in range(5):
            with torch.no_grad():
                if variant.startswith("reference"): torch.cuda.empty_cache()
                _ = model(img, ft, (out, out))
        total = 0.0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if variant == "back to back":
            torch.cuda.synchronize(); s.record()
            with torch.no_grad():
                for _ in range(NUM_RUNS): _ = model(img, ft, (out, out))
            e.record(); torch.cuda.synchronize()
            total = s.elapsed_time(e)
        else:
            for _ in range(NUM_RUNS):
                if variant.startswith("reference"): torch.cuda.empty_cache()
                torch.cuda.synchronize()
                s.record()
                with torch.no_grad():
                    _ = model(img, ft, (out, out))
                e.record(); torch.cuda.synchronize()
                total += s.elapsed_time(e)
        print("%-28s %-50s %.4f ms per forward" % (name, variant, total / NUM_RUNS))
