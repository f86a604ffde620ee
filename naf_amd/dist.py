"""Batch sharding of the NAF forward across the GPUs of one node (one process per GPU, RCCL over xGMI).

The path is embarrassingly parallel over the batch: GroupNorm is per-sample (convolutions.py:83-84)
and RoPE / pooling / attention are per-image, so activations never cross GPUs.  Collectives are used
only off the data path:
  * one broadcast of the flattened parameters (662 528 fp32 = 2.65 MB) at start-up;
  * optional scatter of rank-0-owned inputs;
  * all-gather of small per-rank results (timings, checksums) -- or, on request, of the outputs.
The reference has no distributed code at all; this is new functionality asked
for by BASELINE.json, written directly against torch.distributed ("nccl" == RCCL on ROCm, "gloo" on CPU
for the tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition of ``n_items`` images: first ``n % world`` ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """One flat broadcast of every parameter and persistent buffer (so all ranks run rank-``src``'s
    weights).  2.65 MB for the default NAF: latency-bound, done once."""
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
        off += n


def scatter_batch(full: torch.Tensor | None, shape: Sequence[int], dtype, device, src: int = 0) -> torch.Tensor:
    """Rank ``src`` owns ``full`` [N, ...]; every rank receives its ``shard_range`` slice through ONE collective
    (``torch.distributed.scatter`` = ncclScatter on RCCL: rank ``src`` pushes each slice over its own xGMI link, the
    point-to-point topology's best case).  Slices are padded to the largest shard when N % world != 0 (the collective
    needs equal counts); the pad rows are dropped on arrival."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = int(shape[0])
    spans = [shard_range(n, r, world) for r in range(world)]
    mx = max(b - a for a, b in spans)
    lo, hi = spans[rank]
    if mx == 0:
        return torch.empty((0, *shape[1:]), dtype=dtype, device=device)
    # gloo (CPU tests, and bench.py's dry run with ranks sharing one GPU) scatters host tensors; RCCL device tensors
    via_host = dist.get_backend() == "gloo" and torch.device(device).type != "cpu"
    final_device, device = device, ("cpu" if via_host else device)
    # The owner's argument check is COLLECTIVE (as in broadcast_batch): a one-element status broadcast first, so that a bad `full`
    # raises on every rank instead of leaving the others blocked in the scatter until the process-group timeout.
    # shape AND dtype: a full tensor of another dtype would fail inside dist.scatter on the owner only
    ok = rank != src or (full is not None and tuple(full.shape) == tuple(shape) and full.dtype == dtype)
    status = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.broadcast(status, src=src)
    if int(status.item()) != 1:
        raise ValueError(f"scatter_batch: rank {src} must pass the full tensor of shape {tuple(shape)} and dtype {dtype}")
    recv = torch.empty((mx, *shape[1:]), dtype=dtype, device=device)
    chunks = None
    if rank == src:
        chunks = []
        for a, b in spans:
            if b - a == mx:
                chunks.append(full[a:b].contiguous().to(device))
            else:
                c = torch.zeros((mx, *shape[1:]), dtype=dtype, device=device)
                c[: b - a] = full[a:b]
                chunks.append(c)
    dist.scatter(recv, scatter_list=chunks, src=src)
    return recv[: hi - lo].to(final_device)


def broadcast_batch(full: torch.Tensor | None, shape: Sequence[int], dtype, device, src: int = 0) -> torch.Tensor:
    """Alternative input distribution: ONE ncclBroadcast of the whole batch, every rank keeps its ``shard_range`` slice
    (world x the bytes of ``scatter_batch`` on the wire, but a single ring-pipelined collective; useful when every rank
    needs the whole batch anyway, e.g. for a later gather-free evaluation)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    via_host = dist.get_backend() == "gloo" and torch.device(device).type != "cpu"
    final_device, device = device, ("cpu" if via_host else device)
    # The owner's argument check is COLLECTIVE: a one-element status broadcast first, so that a bad `full` raises on every rank
    # instead of leaving the others blocked in the data broadcast until the process-group timeout.
    ok = rank != src or (full is not None and tuple(full.shape) == tuple(shape) and full.dtype == dtype)
    status = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.broadcast(status, src=src)
    if int(status.item()) != 1:
        raise ValueError(f"broadcast_batch: rank {src} must pass the full tensor of shape {tuple(shape)} and dtype {dtype}")
    if rank == src:
        buf = full.contiguous().to(device)
    else:
        buf = torch.empty(tuple(shape), dtype=dtype, device=device)
    dist.broadcast(buf, src=src)
    lo, hi = shard_range(int(shape[0]), rank, world)
    # a private copy of the slice: a view would keep the whole batch alive on every rank (and alias rank src's input)
    return buf[lo:hi].to(final_device, copy=True)


def gather_scalars(values: Sequence[float], device) -> List[List[float]]:
    """All-gather a short list of floats from every rank (timings, checksums)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [o.tolist() for o in outs]


def gather_outputs(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Optional mode: all-gather the per-rank outputs into the full batch (137 GB at G3 -- far more
    expensive than the compute; off by default)."""
    world = dist.get_world_size()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    pad = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: b - a] for o, (a, b) in zip(outs, sizes)], dim=0)


class ShardedNAF:
    """Runs ``model`` on this rank's slice of a batch, in micro-batches to bound the encoder's activations (bf16 guidance
    at full resolution: 1 GB per image in flight).  ``concat=False`` returns the list of micro-batch outputs instead of
    one concatenated tensor (no second copy of a shard that is tens of GB at G3); ``keep_outputs=False`` drops every
    micro-batch's output as soon as the next one is issued (a consumer that reduces the features on the fly, or a
    benchmark leg whose whole batch would not fit) and returns None."""

    def __init__(self, model: torch.nn.Module, micro_batch: int = 2, concat: bool = True, keep_outputs: bool = True):
        self.model = model
        self.micro_batch = max(1, int(micro_batch))
        self.concat, self.keep_outputs = concat, keep_outputs

    def __call__(self, image: torch.Tensor, features: torch.Tensor, output_size):
        outs = []
        for i in range(0, image.shape[0], self.micro_batch):
            o = self.model(image[i:i + self.micro_batch], features[i:i + self.micro_batch], output_size)
            if self.keep_outputs:
                outs.append(o)
        if not self.keep_outputs:
            return None
        if not self.concat:
            return outs
        return torch.cat(outs, dim=0) if len(outs) != 1 else outs[0]
