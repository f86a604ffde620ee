import sys, torch
sys.path.insert(0, "/root/repo")
from naf_amd import ops
dev = torch.device("cuda:0")
for S in (1024, 448):
    img = torch.rand(1, 3, S, S, device=dev); w0 = torch.randn(128, 3, 1, 1, device=dev) * 0.2; b0 = torch.randn(128, device=dev) * 0.1
    st = ops.new_stats(1, dev)
    def f():
        st.zero_(); ops.stem_conv0(img, w0, b0, None, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(S, "stats-only conv0 1x1 (incl. the 128-byte memset): %.4f ms" % (e0.elapsed_time(e1) / 50))
