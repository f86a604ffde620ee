"""Time naf_stem_wgrad (3x3 and 1x1) at the training point 448^2 (and 1024^2).  NAF_HIP_LIB=<variant> python tools/stem_wgrad_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops
dev = torch.device("cuda:0")
for H in ([int(os.environ['WGRAD_H'])] if os.environ.get('WGRAD_H') else (448, 1024)):
    x = torch.randn(1, H, H, 128, device=dev).to(torch.bfloat16)
    dy = torch.randn(1, H, H, 128, device=dev).to(torch.bfloat16)
    xd = x.double().reshape(1, -1, 8, 16)
    st = ops.stats_from_total(torch.stack([xd.sum((1, 3)), (xd * xd).sum((1, 3))], dim=-1).contiguous())
    gw, gb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    for k in (3, 1):
        for _ in range(3):
            ops.stem_wgrad(dy, x, st, gw, gb, 1e-5, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ops.stem_wgrad(dy, x, st, gw, gb, 1e-5, k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        a = ops.stem_act(x, st, gw, gb, 1e-5, pad=0)
        for _ in range(3):
            ops.stem_wgrad(dy, a, None, None, None, 1e-5, k)
        e0.record()
        for _ in range(20):
            ops.stem_wgrad(dy, a, None, None, None, 1e-5, k)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / 20
        e0.record()
        for _ in range(20):
            ops.stem_act(x, st, gw, gb, 1e-5, pad=0)
        e1.record()
        torch.cuda.synchronize()
        ms3 = e0.elapsed_time(e1) / 20
        print(f"   plain (a materialised): {ms2:.4f} ms + act {ms3:.4f} ms")
        print(f"{os.path.basename(os.environ.get('NAF_HIP_LIB', 'default')):24s} {H}^2 k={k}: {ms:.4f} ms  {2 * H * H * 128 * 128 * k * k / ms / 1e9:.0f} TFLOP/s")
