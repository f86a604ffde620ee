#!/usr/bin/env python
"""NAF forward benchmark on MI355X -- the driver's contract.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one whole NAF forward (conv stem -> RoPE + key pooling -> cross-scale neighbourhood
attention) over one batch of synthetic inputs that are resident in HBM before the timed region.
Workload at N=1 = BASELINE.json configs[1] ("G1"): 1x3x1024x1024 guidance, 1x768x64x64 features ->
1024x1024, window 7, bf16, random-init default-dim weights.  For N>1 every rank runs the same
per-GPU workload on its own images (weak scaling, no data-path collective: the path is
embarrassingly parallel over the batch; RCCL is used for the parameter broadcast and the timing
reduction only).  Rank 0 prints ONE JSON line.

value     = total output pixels of all ranks / max-over-ranks time           [Mpix/s]
roofline  = the attention kernel (naf_xna_fwd, MFMA cell kernel): algorithmic bytes per launch
            (e*B*(256*Ho*Wo + (256+C)*h*w + C*Ho*Wo)) / its mean duration, measured
            with HIP events on the launch stream inside the timed region, vs 8 TB/s.
cpu_baseline = the CPU oracle (oracle/naf_oracle.py, a port -- the reference's NATTEN CPU path is not
            installable) timed on this host's cores on a bounded crop of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (C, lr, out, kernel)
    "G1": (768, 64, 1024, 7),
    "G2-k7": (1024, 32, 512, 7),
    "G2-k11": (1024, 32, 512, 11),
    "G2-k15": (1024, 32, 512, 15),
    "G3": (1024, 64, 1024, 7),
    "G4": (768, 128, 2048, 7),
    # the reference's own published timing point (test/test_results.json:243-256, A100-40GB, fp32):
    # image 448^2, DINO-S features 384 x 28^2 -> 448^2, NAF() default window 9: 56.24 ms = 3.57 Mpix/s
    "REF448": (384, 28, 448, 9),
}
PUBLISHED_MPIX = {"REF448": 3.57}   # BASELINE.md numbers for the exact configuration (other hardware)


class EventTimer:
    """Records HIP event pairs on the current stream around named launches (naf_amd.ops.KERNEL_TIMER)."""

    def __init__(self):
        self.open = {}
        self.pairs = {}
        self.enabled = False

    def start(self, name):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.open[name] = e

    def stop(self, name):
        if self.enabled and name in self.open:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.pairs.setdefault(name, []).append((self.open.pop(name), e))

    def mean_ms(self, name):
        p = self.pairs.get(name)
        if not p:
            return None
        return sum(a.elapsed_time(b) for a, b in p) / len(p)

    def count(self, name):
        return len(self.pairs.get(name, []))


def algorithmic_bytes(B, C, lr, out, elt=2, cq=256):
    return elt * B * (cq * out * out + (cq + C) * lr * lr + C * out * out)


def cpu_baseline(C, ksz, budget_s=20.0):
    """Oracle timed on host cores on a crop of the workload (same channels, same window, 256^2 output
    from a 16^2 feature grid = same ratio 16); cost per output pixel is size-independent."""
    from oracle import naf_oracle as O
    crop, lr = 256, 16
    p = O.make_params(seed=0)
    img = O.hash_normal((1, 3, crop, crop), 1)
    ft = O.hash_normal((1, C, lr, lr), 2)
    with torch.no_grad():
        O.naf_forward_fast(p, img, ft, (crop, crop), kernel_size=ksz)      # warm-up (thread pool, allocator)
        reps, t0 = 0, time.perf_counter()
        while True:
            O.naf_forward_fast(p, img, ft, (crop, crop), kernel_size=ksz)
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s or reps >= 50:
                break
    return {"value": round(crop * crop * reps / el / 1e6, 4), "unit": "Mpix/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{reps} x oracle.naf_forward_fast fp32 on a 1x3x{crop}x{crop} crop, 1x{C}x{lr}x{lr} features, "
                      f"window {ksz} ({el:.1f} s; host has {os.cpu_count()} logical cpus)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="G1", choices=sorted(WORKLOADS))
    ap.add_argument("--per-gpu-batch", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--attention-only", action="store_true", help="time only RoPE'd-Q -> output (scope A)")
    ap.add_argument("--no-fuse-rope", action="store_true", help="materialise the rotated queries (A/B against rotate-on-load)")
    ap.add_argument("--no-fuse-conv0", action="store_true", help="store the 1x1 branch's conv0 activation (A/B against recompute)")
    ap.add_argument("--multi-call", action="store_true", help="one foreign call per kernel (per-kernel phase timers) instead of "
                    "naf_forward's single call")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a hipGraph (NAF.capture); no per-kernel "
                    "event timing is possible inside a graph, so `roofline` is null in this mode")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the NAF hot path has no CPU implementation)")
    # NAF_BENCH_BACKEND=gloo: dry run of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives go through gloo) -- for testing this script only, never for reported numbers
    backend = os.environ.get("NAF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from naf_amd import NAF, ops
    from naf_amd import dist as nd

    C, lr, out, ksz = WORKLOADS[args.workload]
    B = args.per_gpu_batch
    torch.manual_seed(0)                                        # same random-init weights on every rank...
    model = NAF(kernel_size=ksz).to(dev).eval()
    model.fuse_rope = not args.no_fuse_rope
    model.single_call = not args.multi_call
    model.image_encoder.fuse_conv0 = not args.no_fuse_conv0
    if world > 1:
        nd.broadcast_parameters(model, src=0)                   # ...and made identical by one RCCL broadcast
    g = torch.Generator(device=dev).manual_seed(1000 + rank)    # each rank owns different images
    image = torch.randn(B, 3, out, out, device=dev, generator=g)
    feats = torch.randn(B, C, lr, lr, device=dev, generator=g).to(torch.bfloat16)
    size = (out, out)

    timer = EventTimer()
    ops.KERNEL_TIMER = timer

    if args.attention_only:
        q5, k5, tabs = model.guidance_qk(image, feats.shape[-2:], size,
                                         fuse_for=(C // model.upsampler.num_heads, torch.bfloat16))

        def step():
            return model.upsampler(q5, k5, feats, rope_tables=tabs)
    elif args.graph:
        graphed = model.capture(image, feats, size)

        def step():
            return graphed()
    else:
        def step():
            return model(image, feats, size)

    with torch.no_grad():
        for _ in range(args.warmup):
            o = step()
        del o
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        timer.enabled = False

    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        ms_step = el * 1e3 / args.steps
        value = world * B * out * out / (el / args.steps) / 1e6
        xna_ms = timer.mean_ms("xna_mfma")
        alg = algorithmic_bytes(B, C, lr, out)
        roof = None
        if xna_ms:
            ach = alg / (xna_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(args.workload)
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": "xna_mfma_kernel",
                    "kernel_ms": round(xna_ms, 4), "launches": timer.count("xna_mfma"), "algorithmic_bytes": alg,
                    # the same kernel against the matrix pipe (large windows approach the MFMA ridge):
                    # 2 * k^2 * (256 + C) FLOP per output pixel, dense bf16 MFMA peak 2.5 PFLOP/s
                    "mfma_tflops": round(2.0 * ksz * ksz * (256 + C) * B * out * out / (xna_ms * 1e-3) / 1e12, 1),
                    "mfma_frac": round(2.0 * ksz * ksz * (256 + C) * B * out * out / (xna_ms * 1e-3) / 2.5e15, 4)}
        line = {
            "metric": "upsampled Mpixels/sec (NAF forward)", "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (round(value / PUBLISHED_MPIX[args.workload], 2) if (args.workload in PUBLISHED_MPIX and world == 1 and B == 1
                                                                                  and not args.attention_only) else None),
            "dtype": "bf16",
            "data": "synthetic" if backend == "nccl" else f"synthetic (DRY RUN over {backend}, ranks share GPUs: not a measurement)",
            "config": {"workload": f"{args.workload}: {B}x3x{out}x{out} guidance, {B}x{C}x{lr}x{lr} features -> "
                                   f"{out}x{out}, window {ksz}, per GPU", "per_gpu_batch": B, "parallelism": f"batch-shard x{world}",
                       "scope": ("attention-only (scope A)" if args.attention_only else "whole forward (conv stem + RoPE/pool + attention)")
                                + (", hipGraph replay" if args.graph else ""),
                       "weights": "random-init NAF() defaults (dim 256, 4 heads)"},
            "roofline": roof,
            "phases_ms": {k: (round(timer.mean_ms(k), 4) if timer.mean_ms(k) else None)
                          for k in ("stem", "stem_conv0", "stem_conv1", "stem_conv3", "rope_pool", "attention", "xna_mfma")},
            "launches_per_step": {k: timer.count(k) // max(1, args.steps) for k in ("stem_conv0", "stem_conv1", "stem_conv3")},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(C, ksz)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
