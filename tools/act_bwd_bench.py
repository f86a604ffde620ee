"""Time naf_stem_act_bwd (both phases together, as the training step calls it) at 448^2 and 1024^2, plain and with the folded border.
NAF_HIP_LIB=<variant> python tools/act_bwd_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops
dev = torch.device("cuda:0")
for H in (448, 1024):
    x = torch.randn(1, H, H, 128, device=dev).to(torch.bfloat16)
    xd = x.double().reshape(1, -1, 8, 16)
    st = ops.stats_from_total(torch.stack([xd.sum((1, 3)), (xd * xd).sum((1, 3))], dim=-1).contiguous())
    gw, gb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    for fold in (False, True):
        da = torch.randn(1, H + 2 * fold, H + 2 * fold, 128, device=dev).to(torch.bfloat16)
        dx = torch.empty_like(x)
        for _ in range(3):
            ops.stem_act_bwd(da, x, st, gw, gb, 1e-5, dx, fold=fold)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            ops.stem_act_bwd(da, x, st, gw, gb, 1e-5, dx, fold=fold)
        e1.record(); torch.cuda.synchronize()
        print("%-14s act_bwd %d^2 fold=%d: %.4f ms (both phases + the sums' memset)" % (os.path.basename(os.environ.get("NAF_HIP_LIB", "default")), H, fold, e0.elapsed_time(e1) / 20))
