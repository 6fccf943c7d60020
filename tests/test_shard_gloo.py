"""CPU, world_size 2 over gloo: the N>1 frame partition and the scatter/gather exchange."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_range_partitions():
    from desktop2stereo_amd.shard import frame_range, stream_owner
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            blocks = [frame_range(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert [stream_owner(s, 8) for s in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from desktop2stereo_amd.shard import frame_range, gather_outputs, scatter_frames
    H, W = 6, 8
    dev = torch.device("cpu")
    full = None
    if rank == 0:
        full = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n_frames, H, W, 3), dtype=np.uint8))
    mine = scatter_frames(full, n_frames, (H, W, 3), dev)
    lo, hi = frame_range(n_frames, world, rank)
    assert mine.shape[0] == hi - lo
    # stand-in for the per-frame hot path: a deterministic per-frame map (Full-SBS doubles the width)
    out = torch.cat([mine, 255 - mine], dim=2)
    got = gather_outputs(out, n_frames)
    # barrier + MAX-over-ranks timing, as bench.py does
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        ok = bool(torch.equal(got, torch.cat([full, 255 - full], dim=2))) and float(t.item()) == float(world)
        q.put(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 8, 1])
def test_scatter_gather_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_frames) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
