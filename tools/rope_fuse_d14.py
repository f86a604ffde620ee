import sys, torch
sys.path.insert(0, "/root/repo")
import naf_amd
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (lr, out, C, ks) in ((37, 518, 768, 9), (37, 518, 384, 9), (16, 224, 384, 7), (64, 896, 768, 7)):
    m = naf_amd.NAF(kernel_size=ks).to(dev).eval()
    img = torch.randn(1, 3, out, out, device=dev)
    for dt in (torch.bfloat16, torch.float32):
        ft = torch.randn(1, C, lr, lr, device=dev).to(dt)
        r = []
        for fuse in (True, False):
            m.fuse_rope = fuse
            m.__dict__.pop("_plan_cache", None)
            r.append(t(lambda: m(img, ft, (out, out))))
        print("%d -> %d C%d k%d %s: fused %.3f ms  materialised %.3f ms" % (lr, out, C, ks, str(dt)[6:], r[0], r[1]))
