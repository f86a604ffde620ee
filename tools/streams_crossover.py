"""Whole forward with the stem on one stream vs two (NAF_HIP_KNOBS=1 NAF_STEM_STREAMS=1|2; the knob is read once per process), by image
size: python tools/streams_crossover.py S [S ...]  -- prints ms per forward (200 forwards, events on the caller's stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd

dev = torch.device("cuda:0")
torch.manual_seed(0)
naf = naf_amd.NAF().to(dev).eval()
for S in [int(a) for a in sys.argv[1:]] or [512, 768, 1024]:
    lr = S // 16
    img = torch.randn(1, 3, S, S, device=dev)
    ft = torch.randn(1, 768, lr, lr, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        for _ in range(10): naf(img, ft, (S, S))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): naf(img, ft, (S, S))
        e1.record(); torch.cuda.synchronize()
    print("%4d^2  streams=%s  %.4f ms" % (S, os.environ.get("NAF_STEM_STREAMS", "default"), e0.elapsed_time(e1) / 200))
