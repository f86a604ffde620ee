"""A branch's last stem layer with and without the pooled keys riding on it (naf_stem_conv_keys_fwd), interleaved on one lease,
against layer + keys-only pre-pass.  python tools/keys_fuse_time.py [H W] [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import ops

def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 1024)
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(B, H, W, 128, device=dev).to(torch.bfloat16)
    g = x.float().double().view(B, H * W, 8, 16)
    st = ops.stats_from_total(torch.stack([g.sum(dim=(1, 3)), (g * g).sum(dim=(1, 3))], dim=-1).contiguous())
    cat = torch.empty(B, H, W, 256, dtype=torch.bfloat16, device=dev)
    keys = torch.empty(B, H // 16, W // 16, 256, dtype=torch.bfloat16, device=dev)
    gw, gb, bias = torch.ones(128, device=dev), torch.zeros(128, device=dev), torch.zeros(128, device=dev)
    per = torch.logspace(0, 2, 16, device=dev)
    ty, tx = ops.rope_tables(per, H, W)
    catv = cat.permute(0, 3, 1, 2)
    for k in (1, 3):
        wp = (torch.randn(k * k, 128, 128, device=dev) * (0.05 / k)).to(torch.bfloat16)
        br = 0 if k == 1 else 1
        ysl, ksl = cat[..., 128 * br:128 * br + 128], keys[..., 128 * br:128 * br + 128]
        plain = lambda: ops.stem_conv(x, st, gw, gb, 1e-5, wp, bias, ysl, None)
        try:
            fused = lambda: ops.stem_conv(x, st, gw, gb, 1e-5, wp, bias, ysl, None, keys=(ksl, ty, tx))
            fused()
        except Exception as e:
            print(f"{k}x{k}: keys variant not served ({e})")
            fused = None
        for rep in range(3):
            tp = timed(plain)
            tf = timed(fused) if fused else float("nan")
            print(f"last layer {k}x{k} {B}x{H}x{W}: plain {tp:.4f} ms   with keys {tf:.4f} ms   (+{(tf - tp) * 1e3:.1f} us)")
    tpre = timed(lambda: ops.rope_pool(catv, ty, tx, 4, (H // 16, W // 16), write_q=False))
    print(f"keys-only pre-pass over both branches: {tpre:.4f} ms")
main()
