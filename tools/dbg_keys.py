import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import naf_oracle as O
from naf_amd import ops
import test_gpu_keys as T
dev = torch.device("cuda:0")
for (B, H, W) in [(1, 32, 32), (1, 128, 96)]:
    ks = 3
    h, w = H // 16, W // 16
    xd, st_in, gw, gb, wp, bias = T._layer_inputs(dev, B, H, W, ks, 930)
    per = O.rope_periods(256, 4, 100.0)
    ty, tx = ops.rope_tables(per.to(dev), H, W)
    cat1 = torch.zeros((B, H, W, 256), dtype=torch.bfloat16, device=dev)
    keys = torch.full((B, h, w, 256), 7.0, dtype=torch.bfloat16, device=dev)
    ksl = keys[..., 0:128]
    ops.stem_conv(xd, st_in, gw, gb, 1e-5, wp, bias, cat1[..., 0:128], None, keys=(ksl, ty, tx))
    torch.cuda.synchronize()
    y = cat1[..., 0:128].float().cpu().permute(0, 3, 1, 2).contiguous()
    ref = O.key_pool(O.rope(y, per, 2), (h, w))
    got = ksl.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    print(B, H, W, "max err", float(err.max()))
    # per (cell row, 16-channel tile)
    e = err[0].view(8, 16, h, w).amax(dim=1)   # [tile][h][w]
    for t in range(8):
        print(" tile", t, ["%.2e" % float(v) for v in e[t].flatten()])
    # un-rotated sums check: ratio got/ref for a row-angle tile
    import numpy as np
    np.set_printoptions(precision=3, suppress=True, linewidth=200)
    print("got", got[0, :8, 0, 0].numpy(), got[0, 16:24, 0, 0].numpy())
    print("ref", ref[0, :8, 0, 0].numpy(), ref[0, 16:24, 0, 0].numpy())
    # un-rotated mean of the cell for comparison
    um = y[0, :, :16, :16].mean(dim=(1, 2))
    print("unrot mean", um[:8].numpy(), um[16:24].numpy())
    print("ratio got/ref", (got[0, :8, 0, 0] / ref[0, :8, 0, 0]).numpy())
