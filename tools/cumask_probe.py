"""How do CU-masked streams behave?  Times the 3x3 stem layer alone on streams with different CU masks."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import ops, _lib

dev = torch.device("cuda:0")
B, H, W = 1, 1024, 1024
x = torch.randn(B, H, W, 128, device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
st_in = torch.zeros(B, 8, 2, dtype=torch.float64, device=dev); st_in[..., 1] = H * W * 16.0
wp3 = (torch.randn(9, 128, 128, device=dev) * 0.03).to(torch.bfloat16)
wp1 = (torch.randn(1, 128, 128, device=dev) * 0.1).to(torch.bfloat16)
vec = torch.ones(128, device=dev)
lib = _lib.load()

def make(bits, total=256):
    words = (total + 31) // 32
    m = (C.c_uint32 * words)()
    for i in bits: m[i // 32] |= 1 << (i % 32)
    h = C.c_void_p()
    _lib.check(lib.naf_stream_create(C.byref(h), m, words), "create")
    return h

def timeit(h, wp, hint, reps=10):
    for _ in range(2): ops.stem_conv(x, st_in, vec, vec, 1e-5, wp, vec, y, None, stream=h, cu_hint=hint)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ops.stem_conv(x, st_in, vec, vec, 1e-5, wp, vec, y, None, stream=h, cu_hint=hint)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

print("default stream, hint 0      : 3x3 %.3f ms  1x1 %.3f ms" % (timeit(None, wp3, 0), timeit(None, wp1, 0)))
plain = C.c_void_p(); _lib.check(lib.naf_stream_create(C.byref(plain), None, 0), "create")
print("unmasked side stream        : 3x3 %.3f ms  1x1 %.3f ms" % (timeit(plain, wp3, 0), timeit(plain, wp1, 0)))
full = make(range(256))
print("mask all 256                : 3x3 %.3f ms  1x1 %.3f ms" % (timeit(full, wp3, 0), timeit(full, wp1, 0)))
for n in (224, 192, 128):
    pre = make(range(n))
    print("mask prefix [0,%3d) hint %3d: 3x3 %.3f ms  1x1 %.3f ms" % (n, n, timeit(pre, wp3, n), timeit(pre, wp1, n)))
    per = make([i for i in range(256) if (i % 32) < n // 8])
    print("mask %2d per 32-block hint %3d: 3x3 %.3f ms  1x1 %.3f ms" % (n // 8, n, timeit(per, wp3, n), timeit(per, wp1, n)))
for n in (32, 64):
    suf = make(range(256 - n, 256))
    print("mask suffix %3d CUs          : 1x1 %.3f ms" % (n, timeit(suf, wp1, n)))
    per = make([i for i in range(256) if (i % 32) >= 32 - n // 8])
    print("mask last %d per 32-block     : 1x1 %.3f ms" % (n // 8, timeit(per, wp1, n)))
