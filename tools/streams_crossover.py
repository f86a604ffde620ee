"""Whole forward with the stem on one stream vs two (ForwardPlan.streams = 1 | 2: naf_forward_ex's NAF_FWD_ONE_STREAM / NAF_FWD_TWO_STREAMS),
by image size and batch, INTERLEAVED in one process (round 5; round 4 compared two processes):
    python tools/streams_crossover.py [BxS ...]     e.g.  1x512 2x256 1x1024
prints, per geometry, ms per forward for both layouts over three alternating rounds of 100 forwards (events on the caller's stream), the
3x3 layer's launch shape (workgroups x rows) and the library's own plan (naf_forward_streams)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd
from naf_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
naf = naf_amd.NAF(kernel_size=7).to(dev).eval()


def conv3_plan(B, S):
    tiles_x = (S + 31) // 32
    strips = B * tiles_x
    segs = max(1, min(256 // strips if strips <= 256 else 1, (S + 7) // 8))
    seg_h = (-(-S // segs) + 3) // 4 * 4
    return strips * (-(-S // seg_h)), seg_h


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


geoms = [a.split("x") for a in sys.argv[1:]] or [["1", "256"], ["1", "384"], ["1", "448"], ["1", "480"], ["1", "512"], ["1", "544"], ["1", "576"],
                                                  ["1", "640"], ["1", "768"], ["1", "1024"], ["2", "256"], ["4", "256"], ["2", "512"], ["4", "512"],
                                                  ["2", "384"], ["2", "768"]]
for Bs, Ss in geoms:
    B, S = int(Bs), int(Ss)
    lr = S // 16
    img = torch.randn(B, 3, S, S, device=dev)
    ft = torch.randn(B, 768, lr, lr, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        res = {1: [], 2: []}
        ops.ForwardPlan.streams = 0
        naf(img, ft, (S, S))
        plan = naf.__dict__["_plan_cache"][1]
        auto = plan.planned_streams()
        for st in (1, 2):
            ops.ForwardPlan.streams = st
            for _ in range(10):
                naf(img, ft, (S, S))
        torch.cuda.synchronize()
        for rnd in range(3):
            for st in (1, 2):
                ops.ForwardPlan.streams = st
                for _ in range(5):
                    naf(img, ft, (S, S))
                torch.cuda.synchronize()
                res[st].append(timed(lambda: naf(img, ft, (S, S)), 100))
        ops.ForwardPlan.streams = 0
    nb, seg_h = conv3_plan(B, S)
    one, two = min(res[1]), min(res[2])
    print("%dx%4d^2  3x3 launch %3d wg x %3d rows   one %s   two %s   two/one %.3f   library plan: %d" % (
        B, S, nb, seg_h, " ".join("%.4f" % v for v in res[1]), " ".join("%.4f" % v for v in res[2]), two / one, auto), flush=True)
