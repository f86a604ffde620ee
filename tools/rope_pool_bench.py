"""Time the keys-only RoPE + pooling pass on G1's guidance tensor (1 x 256 x 1024 x 1024 bf16 channels-last -> 64 x 64 keys).
    NAF_HIP_LIB=<variant .so> python tools/rope_pool_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(B, 1024, 1024, 256, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
per = torch.logspace(0, 2, 16, device=dev)
ty, tx = ops.rope_tables(per, 1024, 1024)
for _ in range(5):
    ops.rope_pool(x, ty, tx, 4, (64, 64), write_q=False)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
ev[0].record()
N = 50
for _ in range(N):
    ops.rope_pool(x, ty, tx, 4, (64, 64), write_q=False)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / N
print(f"{os.environ.get('NAF_HIP_LIB', 'default'):40s} {ms:.4f} ms  {x.numel() * 2 / ms / 1e6:.0f} GB/s")
