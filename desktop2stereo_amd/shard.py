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


# ---- EMA under frame sharding (second row) ------------------------------------------------------
def ema_step_torch(depth: torch.Tensor, state: torch.Tensor, initialised: bool, alpha: float) -> torch.Tensor:
    """One DepthStabilizer step (reference depth.py:1873-1887) with torch ops, any device: first frame seeds the state and
    passes through; later frames return prev.lerp_(depth, 1 - alpha).  `depth` is overwritten with the returned map.
    (The HIP form is ops.ema_update; this one serves the gloo tests and CPU-side owners.)"""
    if not initialised:
        state.copy_(depth)
        return depth
    state.lerp_(depth, 1.0 - alpha)
    depth.copy_(state)
    return depth


def ema_exchange(depth_local: torch.Tensor, n_frames: int, state: torch.Tensor, initialised: bool, alpha: float,
                 owner: int = 0, ema_step=None) -> Tuple[torch.Tensor, bool]:
    """The temporal EMA (A12) is the one stage of the path that couples frames of ONE stream, and it is a sequential scan in
    frame order.  With a stream's frames block-partitioned over the ranks (frame_range), the post-A11 depth maps -- model
    resolution, 0.6 MB each at 294 x 518 -- travel to the stream's owner, which runs the scan exactly as a single rank would
    (same order, same arithmetic: prev = d0 on the first frame, depth.py:1877-1880) and sends every block back; the 6-50 MB
    frames never move.  Point-to-point like scatter_frames / gather_outputs; no collective.

    depth_local [n_local, h, w] float32: this rank's block, overwritten with the stabilised maps and returned.
    state [h, w] / initialised: the stream's EMA state, meaningful on `owner` only.  Returns (depth_local, initialised')."""
    world, rank = dist.get_world_size(), dist.get_rank()
    step = ema_step or ema_step_torch
    lo, hi = frame_range(n_frames, world, rank)
    assert depth_local.shape[0] == hi - lo, "depth_local must be this rank's frame block"
    if world == 1:
        for i in range(n_frames):
            step(depth_local[i], state, initialised or i > 0, alpha)
        return depth_local, initialised or n_frames > 0
    if rank == owner:
        full = torch.empty((n_frames,) + tuple(depth_local.shape[1:]), dtype=depth_local.dtype, device=depth_local.device)
        reqs = []
        for r in range(world):
            a, b = frame_range(n_frames, world, r)
            if r == owner:
                full[a:b].copy_(depth_local)
            elif b > a:
                reqs.append(dist.irecv(full[a:b], r))
        for q in reqs:
            q.wait()
        for i in range(n_frames):                      # the scan: frame order, one state
            step(full[i], state, initialised or i > 0, alpha)
        reqs = []
        for r in range(world):
            a, b = frame_range(n_frames, world, r)
            if r == owner:
                depth_local.copy_(full[a:b])
            elif b > a:
                reqs.append(dist.isend(full[a:b].contiguous(), r))
        for q in reqs:
            q.wait()
        return depth_local, initialised or n_frames > 0
    if hi > lo:
        dist.send(depth_local.contiguous(), owner)
        dist.recv(depth_local, owner)
    return depth_local, initialised or n_frames > 0
