"""Plan sweep of the table-driven MFMA attention kernel: one process per pinned plan (NAF_UNION_PLAN=ry,seg,dvt)."""
import os, sys, subprocess
os.environ.setdefault("NAF_HIP_KNOBS", "1")   # the A/B knobs below are honoured only with this set
CASES = {"r13.8": (37, 37, 512, 512, 768, 9), "d1k7": (256, 256, 256, 256, 384, 7), "d4": (64, 64, 256, 256, 768, 7), "d8": (64, 64, 512, 512, 768, 7)}
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops
dev = torch.device("cuda:0")
for case in (sys.argv[1:] or list(CASES)):
    h, w, Ho, Wo, C, ks = CASES[case]
    q = torch.randn(1, 4, Ho, Wo, 64, device=dev).to(torch.bfloat16)
    k = torch.randn(1, 4, h, w, 64, device=dev).to(torch.bfloat16)
    v = torch.randn(1, h, w, 4, C // 4, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2, 4)
    o = torch.empty((1, Ho, Wo, 4, C // 4), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2, 4)
    fn = lambda: ops.xna_forward(q, k, v, ks, out=o, path="union")
    res = []
    plans = ["auto"] + ["%d,%d,%d" % (ry, sg, dvt) for ry in (1, 2, 4, 8, 16, 32, 64) for sg in (16, 32, 64, 128, 256) for dvt in (0, C // 8)]
    for pl in plans:
        os.environ.pop("NAF_UNION_PLAN", None)
        if pl != "auto": os.environ["NAF_UNION_PLAN"] = pl
        try:
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 10, pl))
        except Exception as ex:
            pass
    os.environ.pop("NAF_UNION_PLAN", None)
    res.sort()
    print(case, "auto = %.4f ms;" % [t for t, p in res if p == "auto"][0], "best:", "  ".join("%s %.4f" % (p, t) for t, p in res[:6]))
