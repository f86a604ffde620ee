import os, sys, torch
sys.path.insert(0, os.getcwd())
from naf_amd import ops
dev = torch.device("cuda:0")
img = torch.randn(1, 3, 1024, 1024, device=dev)
w = torch.randn(128, 3, 3, 3, device=dev) * 0.2
b = torch.randn(128, device=dev)
y = torch.empty(1, 1024, 1024, 128, dtype=torch.bfloat16, device=dev)
st = ops.new_stats(1, dev)
for _ in range(3): ops.stem_conv0(img, w, b, y, st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): ops.stem_conv0(img, w, b, y, st)
e1.record(); torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(torch.nn.functional.pad(img.double(), (1,1,1,1), mode="reflect"), w.double(), b.double()).permute(0,2,3,1)
err = (y.double() - ref).abs()
rb = ref.to(torch.bfloat16)
mis = (y != rb)
ulp = (rb.double().abs() * 2.0 ** -7).clamp_min(1e-30)          # one bf16 unit in the last place is 2^-8 .. 2^-7 of the value
print(os.environ.get("NAF_CONV0_TERMS", "default"), "terms: %.4f ms;" % (e0.elapsed_time(e1) / 50),
      "outputs that are not bf16(fp64 conv): %.4f %%, of those more than one unit away: %d;" % (100 * float(mis.float().mean()), int(((y.double() - rb.double()).abs() > 1.01 * ulp).sum())),
      "max |y - ref| / (|ref| 2^-8 + 1e-5): %.3f" % float((err / (ref.abs() * 2 ** -8 + 1e-5)).max()))
