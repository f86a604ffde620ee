#!/usr/bin/env python
"""NAF forward benchmark on MI355X -- the driver's contract.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one whole NAF forward (conv stem -> RoPE + key pooling -> cross-scale neighbourhood
attention) over one batch of synthetic inputs that are resident in HBM before the timed region.
Workload at N=1 = BASELINE.json configs[1] ("G1"): 1x3x1024x1024 guidance, 1x768x64x64 features ->
1024x1024, window 7, bf16, random-init default-dim weights.
For N>1 the workload is BASELINE.json configs[3] ("G3"): ONE batch of 64 images of 3x1024x1024 with
1024x64x64 features (DINOv3-ViT-L width) -> 1024x1024, sharded 64/N images per rank (strong scaling: the
total work is fixed).  Rank 0 creates the whole batch; the parameters travel by one flat RCCL broadcast
and the inputs by one RCCL scatter each (naf_amd.dist), both BEFORE the timed region; inside it every rank
runs its shard through naf_amd.dist.ShardedNAF (micro-batches of 8) with no data-path collective -- the
path is embarrassingly parallel over the batch.  After the timed region rank 0 alone runs the same 64
images on its one GPU to report `speedup_vs_1`.  Rank 0 prints ONE JSON line.

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
    # VERDICT r04 item 3's small-image line: 256^2, C = 768, 16^2 features, window 7
    "S256": (768, 16, 256, 7),
    # cells other than 16 x 16 pixels at the reference's timing size: a patch-14 backbone's grid (448 / 14 = 32) and a ratio-8 call
    "P14": (384, 32, 448, 9),
    "R8": (384, 56, 448, 9),
    # the reference's DEFAULT window (NAF() / the released checkpoint: kernel_size 9) at the BASELINE sizes -- not BASELINE configurations
    # (those say window 7), but what a user of hubconf.naf() runs
    "G1-k9": (768, 64, 1024, 9),
    "G3-k9": (1024, 64, 1024, 9),
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


class StepClock:
    """Device time of every timed step: one HIP event on the current stream in front of each step and one
    behind the last -- the forward joins its lent stream back before it returns, so consecutive events bracket whole steps.
    With more than 128 steps only every n-th boundary is recorded (an event record costs ~1 us of host time)."""

    def __init__(self, steps):
        self.stride = max(1, steps // 128)
        self.events = []

    def mark(self, i):
        if i % self.stride == 0:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.append(e)

    def close(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.events.append(e)

    def summary(self, steps):
        """ms per step between consecutive recorded boundaries (after a synchronise)."""
        ev = self.events
        if len(ev) < 2:
            return None
        last = steps - (len(ev) - 2) * self.stride                     # steps inside the last interval
        d = [ev[i].elapsed_time(ev[i + 1]) / (self.stride if i + 2 < len(ev) else max(1, last)) for i in range(len(ev) - 1)]
        srt = sorted(d)
        q = lambda f: srt[min(len(srt) - 1, int(f * len(srt)))]
        return {"steps_per_sample": self.stride, "samples": len(d), "first": round(d[0], 4), "head": [round(x, 4) for x in d[:8]],
                "min": round(srt[0], 4), "median": round(q(0.5), 4), "p90": round(q(0.9), 4), "max": round(srt[-1], 4),
                "sum_ms": round(sum(x * (self.stride if i + 2 < len(ev) else max(1, last)) for i, x in enumerate(d)), 3),
                "what": "device time between HIP events recorded in front of the timed steps (stream order); sum_ms against steps x ms_per_step is the host-side share of the reading"}


def algorithmic_bytes(B, C, lr, out, elt=2, cq=256):
    return elt * B * (cq * out * out + (cq + C) * lr * lr + C * out * out)


def cpu_baseline(C, ksz, lr, out, budget_s=25.0):
    """The CPU oracle (oracle/naf_oracle.py: a port -- the reference's own CPU path needs NATTEN, which cannot be installed)
    timed on this host's cores, SURVEY.md section 8d: (1) BASELINE configs[0] "P1" in full (1x3x224^2, 384x14^2 -> 224^2,
    window 7, fp32), (2) a 256^2 crop of the benched workload (same channels, window and ratio 16: cost per output pixel
    is size-independent), (3) the benched workload in full, once, when (2) predicts it fits the budget.  `value` is (3)
    when it ran, else (2)."""
    from oracle import naf_oracle as O
    cores = torch.get_num_threads()
    p = O.make_params(seed=0)

    def timed(img, ft, size, k, max_s, max_reps):
        with torch.no_grad():
            reps, t0 = 0, time.perf_counter()
            while True:
                O.naf_forward_fast(p, img, ft, size, kernel_size=k)
                reps += 1
                el = time.perf_counter() - t0
                if el > max_s or reps >= max_reps:
                    return reps, el

    with torch.no_grad():
        O.naf_forward_fast(p, O.hash_normal((1, 3, 64, 64), 3), O.hash_normal((1, 64, 4, 4), 4), (64, 64), kernel_size=3)   # thread pool
    res = {"unit": "Mpix/s", "cores": cores, "kind": "port", "host_logical_cpus": os.cpu_count()}
    # (1) P1 in full
    img, ft = O.hash_normal((1, 3, 224, 224), 1), O.hash_normal((1, 384, 14, 14), 2)
    timed(img, ft, (224, 224), 7, 0.0, 1)
    reps, el = timed(img, ft, (224, 224), 7, budget_s * 0.2, 50)
    res["p1_full"] = {"value": round(224 * 224 * reps / el / 1e6, 4), "sample": f"{reps} x P1 (1x3x224x224, 1x384x14x14 -> 224x224, window 7, fp32) in {el:.1f} s"}
    # (2) crop of the benched workload
    crop, clr = 256, 16
    img, ft = O.hash_normal((1, 3, crop, crop), 1), O.hash_normal((1, C, clr, clr), 2)
    timed(img, ft, (crop, crop), ksz, 0.0, 1)
    reps, el = timed(img, ft, (crop, crop), ksz, budget_s * 0.3, 50)
    crop_rate = crop * crop * reps / el / 1e6
    res["crop"] = {"value": round(crop_rate, 4), "sample": f"{reps} x 1x3x{crop}x{crop} crop, 1x{C}x{clr}x{clr} features, window {ksz}, fp32 in {el:.1f} s"}
    res["value"], res["sample"] = res["crop"]["value"], res["crop"]["sample"]
    # (3) the benched workload in full, once
    predicted = out * out / 1e6 / max(crop_rate, 1e-9)
    if predicted <= budget_s * 1.8:
        try:
            img, ft = O.hash_normal((1, 3, out, out), 1), O.hash_normal((1, C, lr, lr), 2)
            reps, el = timed(img, ft, (out, out), ksz, 0.0, 1)
            res["full"] = {"value": round(out * out * reps / el / 1e6, 4),
                           "sample": f"1 x the benched workload in full (1x3x{out}x{out}, 1x{C}x{lr}x{lr} -> {out}x{out}, window {ksz}, fp32) in {el:.1f} s"}
            res["value"], res["sample"] = res["full"]["value"], res["full"]["sample"]
        except (RuntimeError, MemoryError) as e:            # host memory
            res["full"] = {"value": None, "sample": f"failed: {type(e).__name__}"}
    else:
        res["full"] = {"value": None, "sample": f"skipped: predicted {predicted:.0f} s > budget"}
    return res


def measure_traffic_live(workload, per_gpu_batch, timeout_s=90):
    """HBM bytes of ONE launch of the attention kernel, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot
    share a pass: TCC slot limit) of this very script on 3 steps, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE
    tallies 128-byte read requests at 64 bytes: x2).  Returns (bytes, kernel name) or None when rocprofv3 is not available or a
    pass fails -- the caller then falls back to the number committed in profiles/traffic.json and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    vals, kname = {}, None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="naf_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--workload", workload, "--per-gpu-batch", str(per_gpu_batch), "--steps", "3",
                   "--warmup", "1", "--settle-seconds", "0", "--no-cpu-baseline", "--no-live-traffic"]
            # a plain single-GPU child, also when this process is rank 0 of a multi-GPU run (no launcher variables, same device)
            env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                       "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "NAF_BENCH_BACKEND")}
            env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None
            got = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "xna_" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        got.append(float(row["Counter_Value"]))
                        kname = row["Kernel_Name"].split("(")[0][:90]
            if not got:
                return None
            vals[ctr] = sum(got) / len(got)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(vals["WRITE_SIZE"] * 1024 + 2 * vals["FETCH_SIZE"] * 1024), kname


def _self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec this command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`.
    Fails loudly when fewer than N devices are visible (unless NAF_BENCH_BACKEND=gloo: the dry run that shares devices)."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and os.environ.get("NAF_BENCH_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {n} but only {have} ROCm device(s) are visible", file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def _planned_streams(model):
    """Streams the one-call forward used (2 = the encoder branches side by side on a stream this host lends), None outside that mode."""
    plan = (model.__dict__.get("_plan_cache") or (None, None))[1]
    return plan.planned_streams() if plan is not None else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 1000 on one GPU = ~2 s of device time, so that "
                    "a utilisation sampler sees the timed region; 30 on several GPUs, where a step is the whole 64-image batch)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=0, choices=(0, 1, 2), help="stream layout of the one-call forward: 0 = the library's "
                    "plan (naf_forward_streams), 1 = one stream, 2 = the two encoder branches side by side whenever possible (A/B)")
    ap.add_argument("--conv0-exact", action="store_true", help="NAF_FWD_CONV0_EXACT: exact fp32 products in the 3x3 first convolution (A/B)")
    ap.add_argument("--no-cold-reading", action="store_true", help="skip the first reading (W warm-ups + K steps before the settle "
                    "phase, reported as ms_per_step_no_settle)")
    ap.add_argument("--settle-seconds", type=float, default=0.3, help="part of the set-up, before the W warm-up steps: run the step "
                    "for this long so that the device has left its idle power state (a process that starts timing 10 ms after its "
                    "first launch measures the clock ramp: 20 timed steps read 3 %% slower than 1000); 0 disables it")
    ap.add_argument("--phase-every", type=int, default=8, help="record the per-phase events on every n-th timed step")
    ap.add_argument("--no-phase-events", action="store_true", help="do not record the per-phase events inside naf_forward "
                    "(A/B: what the seven extra event records cost)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: G1 on one GPU, G3 on several")
    ap.add_argument("--per-gpu-batch", type=int, default=1, help="single-GPU runs: images per step")
    ap.add_argument("--total-batch", type=int, default=64, help="multi-GPU runs: images of the whole job, sharded over the ranks")
    ap.add_argument("--micro-batch", type=int, default=8, help="multi-GPU runs: images per forward inside a rank's shard "
                    "(G3 on one GPU: 449 / 461 / 465 Mpix/s at 2 / 4 / 8 images per call; 8 images hold 25 GB of activations and output)")
    ap.add_argument("--no-single-gpu-reference", action="store_true", help="multi-GPU runs: skip rank 0's solo run of the whole batch")
    ap.add_argument("--cpu-baseline-budget", type=float, default=25.0, help="seconds of host time for the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc sub-runs "
                    "(two passes of ~15 s after the timed region, one GPU only); use the number committed in profiles/traffic.json")
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
    if args.steps is None:
        args.steps = 1000 if world == 1 else 30
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: start the ranks ourselves -- the same command under torch.distributed.run, one rank
            # per GPU, rendezvous on a free local port; the children print (rank 0: the one JSON line), we return their status
            raise SystemExit(_self_launch(args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the NAF hot path has no CPU implementation)")
    # NAF_BENCH_BACKEND=gloo: dry run of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives go through gloo) -- for testing this script only, never for reported numbers
    backend = os.environ.get("NAF_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())   # a launcher that shows each rank ONE device still works (index 0 everywhere)
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

    if args.workload is None:
        args.workload = "G1" if world == 1 else "G3"
    C, lr, out, ksz = WORKLOADS[args.workload]
    torch.manual_seed(100 + rank)                               # ranks start with DIFFERENT random-init weights...
    model = NAF(kernel_size=ksz).to(dev).eval()
    model.fuse_rope = not args.no_fuse_rope
    model.single_call = not args.multi_call
    model.image_encoder.fuse_conv0 = not args.no_fuse_conv0
    size = (out, out)
    scatter_ms = None
    if world > 1:
        nd.broadcast_parameters(model, src=0)                   # ...made identical by one flat RCCL broadcast (2.65 MB)
        total = args.total_batch
        lo, hi = nd.shard_range(total, rank, world)
        B = hi - lo
        full_img = full_ft = None
        if rank == 0:                                           # rank 0 owns the batch (north_star: "batched images shard ... via RCCL")
            g = torch.Generator(device=dev).manual_seed(1000)
            full_img = torch.randn(total, 3, out, out, device=dev, generator=g).to(torch.bfloat16)
            full_ft = torch.randn(total, C, lr, lr, device=dev, generator=g).to(torch.bfloat16)
        torch.cuda.synchronize()
        dist.barrier()
        t_sc = time.perf_counter()
        image = nd.scatter_batch(full_img, (total, 3, out, out), torch.bfloat16, dev, src=0)
        feats = nd.scatter_batch(full_ft, (total, C, lr, lr), torch.bfloat16, dev, src=0)
        torch.cuda.synchronize()
        dist.barrier()
        scatter_ms = (time.perf_counter() - t_sc) * 1e3
    else:
        B = args.per_gpu_batch
        total = B
        g = torch.Generator(device=dev).manual_seed(1000)
        image = torch.randn(B, 3, out, out, device=dev, generator=g)
        feats = torch.randn(B, C, lr, lr, device=dev, generator=g).to(torch.bfloat16)

    ops.ForwardPlan.streams = args.streams
    ops.ForwardPlan.conv0_exact = bool(args.conv0_exact)
    timer = EventTimer()
    # phase events on every 8th step: seven extra event records between the kernels of a step cost 0.03-0.05 ms (1.5-2 %) of a
    # 2.25 ms G1 step when taken every step (gpurun r9b: 2.270 vs 2.228 ms), a quarter of a percent this way
    timer.phases = 0 if args.no_phase_events else args.phase_every
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
    elif world > 1:
        sharded = nd.ShardedNAF(model, micro_batch=args.micro_batch, concat=False)

        def step():
            return sharded(image, feats, size)      # list of micro-batch outputs: the rank's shard stays resident
    else:
        def step():
            return model(image, feats, size)

    with torch.no_grad():
        o = None
        # Reading 1 (SURVEY 8d / test/forward_speed.py:31-52, the protocol of rounds 1-3): W warm-up steps from idle, then the same
        # --steps timed the same way, BEFORE any settle phase -- reported as ms_per_step_no_settle so that driver numbers stay
        # comparable across rounds.  Reading 2 (ms_per_step, `value`): after the settle phase below.
        el_cold = clock_cold = None
        if args.settle_seconds > 0 and not args.no_cold_reading:
            for _ in range(args.warmup):
                o = step()
            del o
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            clock_cold = StepClock(args.steps)
            t0c = time.perf_counter()
            for i in range(args.steps):
                clock_cold.mark(i)
                o = step()
            clock_cold.close()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el_cold = time.perf_counter() - t0c
            if world > 1:
                tc = torch.tensor([el_cold], dtype=torch.float64, device=dev)
                dist.all_reduce(tc, op=dist.ReduceOp.MAX)
                el_cold = float(tc.item())
        settle_steps = 0
        if args.settle_seconds > 0:      # set-up, like the plan / table / workspace creation of the first call: not a timed or counted step
            t_settle = time.perf_counter()
            while time.perf_counter() - t_settle < args.settle_seconds:
                o = step()
                settle_steps += 1
                if settle_steps % 8 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            o = step()
        del o
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timer.enabled = True
        clock = StepClock(args.steps)
        t0 = time.perf_counter()
        for i in range(args.steps):
            clock.mark(i)
            o = step()
        clock.close()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        timer.enabled = False

    ranks_info = weak = None
    if world > 1:
        my_el = el
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        # Proof that the job ran on `world` DISTINCT GPUs: every rank reports the device it computed on (UUID / PCI bus id
        # from the driver) and its own time for the timed region; under RCCL the identities must differ.
        prop = torch.cuda.get_device_properties(dev)
        ident = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": prop.name,
                 "uuid": str(getattr(prop, "uuid", "")) or None,
                 "pci": (f"{getattr(prop, 'pci_domain_id', 0):04x}:{getattr(prop, 'pci_bus_id', -1):02x}:{getattr(prop, 'pci_device_id', 0):02x}"
                         if hasattr(prop, "pci_bus_id") else None),
                 "images": B, "ms_per_step": round(my_el * 1e3 / args.steps, 3)}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, ident)
        if backend == "nccl":
            # identity = what the driver reports (UUID, PCI bus id) plus the device index the rank selected: a driver that hands
            # out a placeholder UUID to every device must not fail a correctly launched job
            ids = [(r["uuid"], r["pci"], r["device_index"]) for r in ranks_info]
            if len(set(ids)) != world:
                raise SystemExit(f"bench.py: {world} ranks on {len(set(ids))} distinct GPUs: {ids}")
        # Weak-scaling leg beside the strong one: every rank runs `micro_batch` images (one micro-batch, the same per-GPU work
        # at every N) between barriers; afterwards rank 0 runs the same alone.  Raw times only.
        # the same image count on EVERY rank (uneven shards: the smallest one), agreed collectively so that all ranks take the same
        # branch around the barriers below (a rank with an empty shard would otherwise skip them and hang the others)
        wbt = torch.tensor([min(args.micro_batch, B)], dtype=torch.int64, device=dev)
        dist.all_reduce(wbt, op=dist.ReduceOp.MIN)
        wb = int(wbt.item())
        if wb > 0:
            with torch.no_grad():
                for rep in range(2):
                    torch.cuda.synchronize()
                    dist.barrier()
                    tw = time.perf_counter()
                    for _ in range(3):
                        ow = model(image[:wb], feats[:wb], size)
                    torch.cuda.synchronize()
                    dist.barrier()
                    w_all = (time.perf_counter() - tw) / 3
                    del ow
                w_solo = 0.0
                if rank == 0:
                    torch.cuda.synchronize()
                    tw = time.perf_counter()
                    for _ in range(3):
                        ow = model(image[:wb], feats[:wb], size)
                    torch.cuda.synchronize()
                    w_solo = (time.perf_counter() - tw) / 3
                    del ow
                dist.barrier()
            tt = torch.tensor([w_all], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            weak = {"images_per_gpu": wb, "ms_all_ranks_busy": round(float(tt.item()) * 1e3, 3),
                    "ms_rank0_alone": round(w_solo * 1e3, 3) if rank == 0 else None,
                    "mpix_per_s_all_ranks": round(world * wb * out * out / float(tt.item()) / 1e6, 2)}

    one_gpu_ms = None
    if world > 1 and not args.no_single_gpu_reference:
        # the same `total` images on ONE GPU (rank 0 alone, outputs dropped as it goes: 64 outputs are 137 GB)
        del o
        torch.cuda.empty_cache()
        if rank == 0:
            solo = nd.ShardedNAF(model, micro_batch=args.micro_batch, keep_outputs=False)
            with torch.no_grad():
                solo(full_img[: 2 * args.micro_batch], full_ft[: 2 * args.micro_batch], size)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                solo(full_img, full_ft, size)
                torch.cuda.synchronize()
                one_gpu_ms = (time.perf_counter() - t1) * 1e3
        dist.barrier()

    if rank == 0:
        ms_step = el * 1e3 / args.steps
        value = total * out * out / (el / args.steps) / 1e6
        xna_ms = timer.mean_ms("xna_mfma")
        mb = min(args.micro_batch, B) if world > 1 else B          # images per attention launch
        alg = algorithmic_bytes(mb, C, lr, out)
        roof = None
        if xna_ms:
            ach = alg / (xna_ms * 1e-3) / 1e9
            traffic, kname, tsrc = None, "xna_mfma_kernel", None
            # one GPU: this workload; several: rank 0 measures one micro-batch launch of its shard the same way (the other ranks wait
            # at the closing barrier), so that the N > 1 line carries a measured number too, not a constant x micro-batch
            if not args.no_live_traffic and not args.attention_only and not args.graph and backend == "nccl":
                live = measure_traffic_live(args.workload, mb, timeout_s=90 if world == 1 else 240)
                if live is not None:
                    traffic, kname = live
                    tsrc = "measured in this run: rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE in separate passes, mean per launch of the attention kernel"
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if traffic is None and os.path.exists(tpath):
                try:
                    ent = json.load(open(tpath)).get(args.workload)
                    if isinstance(ent, dict):        # PMC bytes of ONE image's launch (profiles/r03_pmc_hbm_traffic.txt) x images per launch
                        traffic, kname = int(ent["bytes"]) * mb, ent.get("kernel", kname)
                    elif ent is not None:
                        traffic = int(ent) * mb
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": tsrc or ("committed constant x images per launch: rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE), profiles/traffic.json" if traffic else None),
                    "kernel": kname,
                    "kernel_ms": round(xna_ms, 4), "launches": timer.count("xna_mfma"), "algorithmic_bytes": alg,
                    # the same kernel against the matrix pipe (large windows approach the MFMA ridge):
                    # 2 * k^2 * (256 + C) FLOP per output pixel, dense bf16 MFMA peak 2.5 PFLOP/s
                    "mfma_tflops": round(2.0 * ksz * ksz * (256 + C) * mb * out * out / (xna_ms * 1e-3) / 1e12, 1),
                    "mfma_frac": round(2.0 * ksz * ksz * (256 + C) * mb * out * out / (xna_ms * 1e-3) / 2.5e15, 4)}
        m = lambda k: timer.mean_ms(k)
        r4 = lambda v: round(v, 4) if v else None
        phases = {k: r4(m(k)) for k in ("stem", "stem_conv0", "stem_conv1", "stem_conv3", "rope_pool", "attention", "xna_mfma")}
        if m("stem_layer_3x3"):
            # single-call mode: naf_forward records events at its phase boundaries (naf_forward_args.phase_events, version >= 105).
            # The two branches' layers alternate (first convolutions, then block layer 0 of both branches, layer 1 of both, ...),
            # so the call brackets ONE launch of each layer kernel (stage 1): stem_conv1 / stem_conv3 are measured per-launch
            # times, not phase / nlayer; the remainder of the stem is the other stages.  rope_pool includes the 9 us value packing.
            nl = 2 * len(list(model.image_encoder.encoder)[1:])
            two = not m("stem_layer_1x1")   # version >= 200: the branches run on two streams, only the 3x3 launch is bracketed
            phases.update({"stem_first_convs": r4(m("stem_first_convs")), "stem_conv1": r4(m("stem_layer_1x1")), "stem_conv3": r4(m("stem_layer_3x3")),
                           "layers_per_branch": nl,
                           "stem_order": ("two streams: the 1x1 branch's layers run beside the 3x3 branch's (stem_conv3 = one 3x3 launch with 1x1 launches beside it)"
                                          if two else "layers of the two branches alternate (1x1, 3x3, 1x1, ...)"),
                           "source": f"hipEvents recorded inside the one naf_forward call, on {timer.count('stem')} of the {args.steps} timed steps"})
        if roof and m("rope_pool"):
            # a separate RoPE / key-pooling pass is overhead against the achieved fraction, not algorithmic traffic
            t_pre = (xna_ms + m("rope_pool")) * 1e-3
            roof["prepass_ms"] = r4(m("rope_pool"))
            roof["frac_with_prepass"] = round(alg / t_pre / 1e9 / HBM_PEAK_GBS, 4)
        line = {
            "metric": "upsampled Mpixels/sec (NAF forward)", "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            # the same W warm-ups + K steps taken BEFORE the settle phase (the protocol of rounds 1-3): comparable across rounds
            "ms_per_step_no_settle": round(el_cold * 1e3 / args.steps, 4) if el_cold else None,
            # where the time of each reading went, step by step (a clock ramp shows as a falling `head`, a slow first launch as `first`)
            "step_ms": clock.summary(args.steps),
            "step_ms_no_settle": clock_cold.summary(args.steps) if clock_cold else None,
            "settle": {"seconds": args.settle_seconds, "steps": settle_steps,
                       "what": "untimed set-up before the warm-up steps: the step is run for this long so that the device has left its idle power state"},
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": (round(value / PUBLISHED_MPIX[args.workload], 2) if (args.workload in PUBLISHED_MPIX and world == 1 and B == 1
                                                                                  and not args.attention_only) else None),
            "dtype": "bf16",
            "data": "synthetic" if backend == "nccl" else f"synthetic (DRY RUN over {backend}, ranks share GPUs: not a measurement)",
            "config": {"workload": (f"{args.workload}: {B}x3x{out}x{out} guidance, {B}x{C}x{lr}x{lr} features -> {out}x{out}, window {ksz}, per GPU"
                                    if world == 1 else
                                    f"{args.workload}: batch of {total} x 3x{out}x{out} guidance, {total} x {C}x{lr}x{lr} features -> {out}x{out}, "
                                    f"window {ksz}, {B} images per GPU in micro-batches of {args.micro_batch}"),
                       "per_gpu_batch": B, "global_batch": total, "parallelism": f"batch-shard x{world}",
                       "input_distribution": (None if world == 1 else "rank 0 -> RCCL scatter (naf_amd.dist.scatter_batch), before the timed region"),
                       "scope": ("attention-only (scope A)" if args.attention_only else "whole forward (conv stem + RoPE/pool + attention)")
                                + (", hipGraph replay" if args.graph else ""),
                       "weights": "random-init NAF() defaults (dim 256, 4 heads)",
                       "streams": _planned_streams(model),
                       # a driver that divides value(N) by value(1) compares two workloads: name them
                       "scale_note": ("N = 1 times G1 (1 image, C = 768: the configuration BASELINE.json's metric is quoted on); the N > 1 lines time "
                                      "G3 (64 images, C = 1024, ~6 % fewer Mpix/s per GPU) and carry the SAME 64 images on one GPU as one_gpu_ms / "
                                      "speedup_vs_1 -- that is the like-for-like scaling figure (or run --gpus 1 --workload G3 --per-gpu-batch 8)"
                                      if world == 1 else
                                      "strong scaling of G3's 64 images; speedup_vs_1 = one_gpu_ms / ms_per_step compares with the same 64 images on "
                                      "rank 0 alone and is the like-for-like figure: the N = 1 line of this script times another workload (G1: 1 image, C = 768)")},
            "roofline": roof,
            "phases_ms": {k: v for k, v in phases.items() if v is not None},
        }
        lps = {k: timer.count(k) // max(1, args.steps) for k in ("stem_conv0", "stem_conv1", "stem_conv3")}
        if any(lps.values()):    # only the composed (multi-call) modes time every launch on its own
            line["launches_per_step"] = lps
        if world > 1:
            line["ranks"] = ranks_info
            line["weak_leg"] = weak
            line["scatter_ms"] = round(scatter_ms, 3)
            line["one_gpu_ms"] = round(one_gpu_ms, 3) if one_gpu_ms else None
            line["speedup_vs_1"] = round(one_gpu_ms / ms_step, 3) if one_gpu_ms else None
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(C, ksz, lr, out, budget_s=args.cpu_baseline_budget)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
