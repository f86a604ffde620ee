"""Multi-process tests of the batch-shard driver on CPU (gloo, world_size 2): partitioning, parameter
broadcast, input scatter, result gathers.  The kernels themselves are GPU-only; here the per-rank
"model" is a deterministic stand-in so that the distributed logic is what is under test."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from naf_amd import dist as nd


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [nd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        nd.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
        from naf_amd import NAF
        model = NAF(dim=32, heads_attn=2, heads_rope=2, kernel_size=3)
        nd.broadcast_parameters(model, src=0)
        flat = torch.cat([p.reshape(-1) for p in model.parameters()] + [model.image_encoder.rope.periods])
        sums = nd.gather_scalars([float(flat.double().sum()), float(flat.double().abs().sum())], "cpu")
        assert all(s == sums[0] for s in sums), "parameters differ after broadcast"

        N = 5
        full_img = torch.arange(N * 3 * 4 * 4, dtype=torch.float32).view(N, 3, 4, 4) if rank == 0 else None
        img = nd.scatter_batch(full_img, (N, 3, 4, 4), torch.float32, "cpu", src=0)
        lo, hi = nd.shard_range(N, rank, world)
        expect = torch.arange(N * 3 * 4 * 4, dtype=torch.float32).view(N, 3, 4, 4)[lo:hi]
        assert torch.equal(img, expect)
        img_b = nd.broadcast_batch(full_img, (N, 3, 4, 4), torch.float32, "cpu", src=0)
        assert torch.equal(img_b, expect)
        # a private slice: not a view of the whole batch (which would stay alive on every rank), not rank 0's input
        assert img_b.untyped_storage().nbytes() == img_b.numel() * 4
        if rank == 0:
            assert img_b.data_ptr() != full_img.data_ptr()
        # missing / wrong shape / wrong dtype on the owner: the check is collective -- EVERY rank enters the call and every rank
        # raises (a src-only raise would leave the others blocked in the broadcast until the process-group timeout)
        for which in range(3):
            bad = None if rank != 0 else (None, full_img[:-1], full_img.double())[which]
            with pytest.raises(ValueError):
                nd.broadcast_batch(bad, (N, 3, 4, 4), torch.float32, "cpu", src=0)
        for which in range(3):                               # the same for the scatter (round 5: collective; round 6: the dtype too)
            bad = None if rank != 0 else (None, full_img[:-1], full_img.double())[which]
            with pytest.raises(ValueError):
                nd.scatter_batch(bad, (N, 3, 4, 4), torch.float32, "cpu", src=0)
        empty = nd.scatter_batch(torch.zeros(0, 2) if rank == 0 else None, (0, 2), torch.float32, "cpu", src=0)
        assert empty.shape == (0, 2)

        class Fake(torch.nn.Module):                         # per-image op: no cross-sample term
            def forward(self, image, feats, size):
                return image.mean(dim=(1, 2, 3), keepdim=True) + feats
        feats = torch.ones(hi - lo, 1, 1, 1) * (rank + 1)
        out = nd.ShardedNAF(Fake(), micro_batch=2)(img, feats, (4, 4))
        assert out.shape[0] == hi - lo
        parts = nd.ShardedNAF(Fake(), micro_batch=2, concat=False)(img, feats, (4, 4))
        assert isinstance(parts, list) and torch.equal(torch.cat(parts), out)
        assert nd.ShardedNAF(Fake(), micro_batch=2, keep_outputs=False)(img, feats, (4, 4)) is None
        allout = nd.gather_outputs(out, N)
        ref = torch.arange(N * 48, dtype=torch.float32).view(N, 48).mean(dim=1).view(N, 1, 1, 1)
        owner = torch.tensor([1.0 if i < nd.shard_range(N, 0, world)[1] else 2.0 for i in range(N)]).view(N, 1, 1, 1)
        assert torch.allclose(allout, ref + owner)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_roundtrip():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
