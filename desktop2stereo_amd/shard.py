"""Frame-level data parallelism: independent frames shard across the GPUs of a node.

The reference is strictly one device, batch 1, no collectives.  Frames are
independent (except the optional EMA, which is per stream), so the partition needs NO data-path
collective: each rank owns a contiguous block of the frame index space and a full model replica
(weights <= 0.7 GB).  For an ingest-on-rank-0 deployment the only exchange is the trivial
scatter of uint8 frames / gather of packed stereo frames (6-50 MB each, ~40 us/link over xGMI),
implemented here over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def frame_range(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank; blocks differ by at most one frame."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def stream_owner(stream_id: int, world: int) -> int:
    """Stateful streams (EMA / video models) never split: stream s lives on rank s % world."""
    return stream_id % world


def scatter_frames(frames: Optional[torch.Tensor], n_frames: int, shape: Tuple[int, int, int],
                   device: torch.device, src: int = 0) -> torch.Tensor:
    """Rank `src` holds uint8 [n_frames,H,W,3]; every rank returns its own block.
    Point-to-point sends (xGMI is point-to-point; a root-centric scatter is bounded by the root's
    egress, ~1 TB/s = 170k 1080p frames/s): no ring, no all-to-all."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = frame_range(n_frames, world, rank)
    mine = torch.empty((hi - lo,) + tuple(shape), dtype=torch.uint8, device=device)
    if rank == src:
        reqs = []
        for r in range(world):
            a, b = frame_range(n_frames, world, r)
            if r == src:
                mine.copy_(frames[a:b])
            elif b > a:
                reqs.append(dist.isend(frames[a:b].contiguous(), r))
        for q in reqs:
            q.wait()
    elif hi > lo:
        dist.recv(mine, src)
    return mine


def gather_outputs(out: torch.Tensor, n_frames: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Inverse of scatter_frames for the packed stereo frames; returns the full stack on `dst`."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst:
        full = torch.empty((n_frames,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
        reqs = []
        for r in range(world):
            a, b = frame_range(n_frames, world, r)
            if r == dst:
                full[a:b].copy_(out)
            elif b > a:
                reqs.append(dist.irecv(full[a:b], r))
        for q in reqs:
            q.wait()
        return full
    if out.shape[0] > 0:
        dist.send(out.contiguous(), dst)
    return None
