"""Frame-level data parallelism: independent frames shard across the GPUs of a node.

The reference is strictly one device, batch 1, no collectives.  Frames are
independent (except the optional EMA, which is per stream), so the partition needs NO data-path
collective: each rank owns a contiguous block of the frame index space and a full model replica
(weights <= 0.7 GB).  For an ingest-on-rank-0 deployment the only exchange is the trivial
scatter of uint8 frames / gather of packed stereo frames (6-50 MB each over xGMI),
implemented here over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

Round 4: every exchange is ONE grouped point-to-point batch (dist.batch_isend_irecv: RCCL launches the sends / receives of a
group together, so the root's seven xGMI links move their blocks concurrently instead of one peer after another), and
PipelinedIngest overlaps the scatter of step i + 1 and the gather of step i - 1 with the compute of step i (side streams,
double buffers).  Link budget, rank-0 ingest on an 8-GPU MI355X node (xGMI is point-to-point: 7 links per GPU, ~50-64 GB/s per
direction each; a root-centric exchange is bounded by the root's links, all seven usable at once because every peer has its
own link):   1080p, Full-SBS: 6.2 MB in + 12.4 MB out per frame.
    batch  1 per GPU: 7 x 6.2 MB out of the root, 7 x 12.4 MB into it per step -> 0.10-0.12 ms / 0.19-0.25 ms on the wire per
             ~1.14 ms compute step: hidden completely once overlapped;
    batch 32 per GPU: 7 x 199 MB out, 7 x 398 MB in per ~9.2 ms step -> 3.1-4.0 ms / 6.2-8.0 ms on the wire: the inbound
             direction uses 70-85 % of a step -- it only fits because it is overlapped, and it is the first thing that stops
             scaling (a capture host that wants > ~28 k frames/s of Full-SBS output back on one GPU needs a second ingest rank).
No all-reduce or ring anywhere: the 7-link ring bound of collectives does not apply.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

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


class _Pending:
    """Handle of one grouped exchange: wait() blocks until every send / receive of the group has completed (for RCCL: until the
    current stream is ordered behind them -- no host synchronisation)."""

    def __init__(self, reqs, keep=()):
        self.reqs = list(reqs)
        self.keep = keep                      # tensors the in-flight operations read (contiguous copies): alive until wait()

    def wait(self):
        for q in self.reqs:
            q.wait()
        self.reqs, self.keep = [], ()


def _launch(ops: List[dist.P2POp]) -> list:
    return dist.batch_isend_irecv(ops) if ops else []


def scatter_frames(frames: Optional[torch.Tensor], n_frames: int, shape: Tuple[int, int, int],
                   device: torch.device, src: int = 0, out: Optional[torch.Tensor] = None, async_op: bool = False):
    """Rank `src` holds uint8 [n_frames,H,W,3]; every rank returns its own block.  One grouped batch of point-to-point sends
    from the root (no ring, no all-to-all).  async_op: returns (block, handle); the block is valid after handle.wait()."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = frame_range(n_frames, world, rank)
    mine = out if out is not None else torch.empty((hi - lo,) + tuple(shape), dtype=torch.uint8, device=device)
    assert mine.shape[0] == hi - lo
    ops, keep = [], []
    if rank == src:
        for r in range(world):
            a, b = frame_range(n_frames, world, r)
            if r == src:
                mine.copy_(frames[a:b])
            elif b > a:
                blk = frames[a:b].contiguous()
                keep.append(blk)
                ops.append(dist.P2POp(dist.isend, blk, r))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, mine, src))
    h = _Pending(_launch(ops), keep)
    if async_op:
        return mine, h
    h.wait()
    return mine


def gather_outputs(out: torch.Tensor, n_frames: int, dst: int = 0, full: Optional[torch.Tensor] = None, async_op: bool = False):
    """Inverse of scatter_frames for the packed stereo frames; returns the full stack on `dst` (None elsewhere).  One grouped
    batch.  async_op: returns (full | None, handle)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ops, keep = [], []
    if rank == dst:
        if full is None:
            full = torch.empty((n_frames,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
        for r in range(world):
            a, b = frame_range(n_frames, world, r)
            if r == dst:
                full[a:b].copy_(out)
            elif b > a:
                ops.append(dist.P2POp(dist.irecv, full[a:b], r))
    else:
        full = None
        if out.shape[0] > 0:
            blk = out.contiguous()
            keep.append(blk)
            ops.append(dist.P2POp(dist.isend, blk, dst))
    h = _Pending(_launch(ops), keep)
    if async_op:
        return full, h
    h.wait()
    return full


# ---- EMA under frame sharding (second row) ------------------------------------------------------
def _ema_step_hip(depth: torch.Tensor, state: torch.Tensor, initialised: bool, alpha: float) -> torch.Tensor:
    """One DepthStabilizer step (reference depth.py:1873-1887) through d2s_ema_update (post.hip): device tensors only."""
    from . import ops
    return ops.ema_update(depth, state, initialised, alpha)


def ema_exchange(depth_local: torch.Tensor, n_frames: int, state: torch.Tensor, initialised: bool, alpha: float,
                 owner: int = 0, ema_step: Optional[Callable] = None) -> Tuple[torch.Tensor, bool]:
    """The temporal EMA (A12) is the one stage of the path that couples frames of ONE stream, and it is a sequential scan in
    frame order.  With a stream's frames block-partitioned over the ranks (frame_range), the post-A11 depth maps -- model
    resolution, 0.6 MB each at 294 x 518 -- travel to the stream's owner, which runs the scan exactly as a single rank would
    (same order, same arithmetic: prev = d0 on the first frame, depth.py:1877-1880) and sends every block back; the 6-50 MB
    frames never move.  Two grouped point-to-point batches (in, out); no collective.

    depth_local [n_local, h, w] float32: this rank's block, overwritten with the stabilised maps and returned.
    state [h, w] / initialised: the stream's EMA state, meaningful on `owner` only.  Returns (depth_local, initialised').
    ema_step(depth, state, initialised, alpha): the scan's step; default = the HIP kernel (d2s_ema_update) -- device tensors.
    There is no CPU arithmetic in this package: host tensors (the gloo tests) must bring their own step."""
    step = ema_step
    if step is None:
        if not depth_local.is_cuda:
            raise RuntimeError("ema_exchange: host tensors need an explicit ema_step (the package computes on the GPU only)")
        step = _ema_step_hip
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
    lo, hi = frame_range(n_frames, world, rank)
    assert depth_local.shape[0] == hi - lo, "depth_local must be this rank's frame block"
    if world == 1:
        for i in range(n_frames):
            step(depth_local[i], state, initialised or i > 0, alpha)
        return depth_local, initialised or n_frames > 0
    if rank == owner:
        full = torch.empty((n_frames,) + tuple(depth_local.shape[1:]), dtype=depth_local.dtype, device=depth_local.device)
        ops = []
        for r in range(world):
            a, b = frame_range(n_frames, world, r)
            if r == owner:
                full[a:b].copy_(depth_local)
            elif b > a:
                ops.append(dist.P2POp(dist.irecv, full[a:b], r))
        _Pending(_launch(ops)).wait()
        for i in range(n_frames):                      # the scan: frame order, one state
            step(full[i], state, initialised or i > 0, alpha)
        ops = []
        for r in range(world):
            a, b = frame_range(n_frames, world, r)
            if r == owner:
                depth_local.copy_(full[a:b])
            elif b > a:
                ops.append(dist.P2POp(dist.isend, full[a:b], r))
        _Pending(_launch(ops), (full,)).wait()
        return depth_local, initialised or n_frames > 0
    if hi > lo:
        blk = depth_local.contiguous()
        _Pending(_launch([dist.P2POp(dist.isend, blk, owner)]), (blk,)).wait()
        _Pending(_launch([dist.P2POp(dist.irecv, depth_local, owner)])).wait()
    return depth_local, initialised or n_frames > 0


# ---- one stream, frames sharded, temporal smoothing on -----------------------------------------------------------------
class ShardedStream:
    """predict_depth(use_temporal_smooth=True) + make_sbs for ONE stream whose frames are block-partitioned over the ranks
    (the sharded counterpart of depth.pipeline(..., use_temporal_smooth=True)):

        frames_local -> pre-process + model + post-process   (independent frames, this rank's block)
                     -> ema_exchange                          (the stream's scan, on its owner, in frame order)
                     -> warp with the stabilised maps         (independent frames again)

    stages: an object with  depth_small(frames_u8[n,H,W,3]) -> float32 [n,h,w]  (A2-A11) and  warp(frames_u8, depth_small) ->
    packed frames; the default binds the HIP stages of an ops.Engine.  The EMA state lives on the stream's owner."""

    def __init__(self, stages, stream_id: int = 0, alpha: float = 0.9, ema_step: Optional[Callable] = None):
        self.stages, self.alpha, self.ema_step = stages, alpha, ema_step
        self.owner = stream_owner(stream_id, dist.get_world_size() if dist.is_initialized() else 1)
        self.state: Optional[torch.Tensor] = None
        self.initialised = False

    def reset(self):
        self.initialised = False

    def __call__(self, frames_local: torch.Tensor, n_frames: int) -> torch.Tensor:
        d = self.stages.depth_small(frames_local)
        if self.state is None or self.state.shape != d.shape[1:] or self.state.device != d.device:
            self.state = torch.zeros(d.shape[1:], dtype=d.dtype, device=d.device)
            self.initialised = False
        d, self.initialised = ema_exchange(d, n_frames, self.state, self.initialised, self.alpha, self.owner, self.ema_step)
        return self.stages.warp(frames_local, d)


class EngineStages:
    """The HIP stages of one ops.Engine in the shape ShardedStream wants."""

    def __init__(self, engine, params, sbs_params, out_fmt=None):
        from . import _lib, ops
        self.ops, self.eng, self.p, self.sp = ops, engine, params, sbs_params
        self.fmt = _lib.FMT_U8_HWC if out_fmt is None else out_fmt

    def depth_small(self, frames):
        p = self.p
        x = self.ops.preprocess(frames, p.depth_resolution, self.eng.cfg.patch, p.mean, p.std, p.resample, p.square_input)
        return self.ops.post_process_depth(self.eng(x), p)

    def warp(self, frames, depth_small):
        return self.ops.make_sbs(frames, depth_small, self.sp, self.fmt)


# ---- rank-0 ingest, software-pipelined -----------------------------------------------------------------------------------
class PipelinedIngest:
    """Rank-0 ingest as a three-stage software pipeline over steps (double buffers on every rank):

        call k:   scatter(k) starts  |  compute(k - 1)  |  gather(k - 2) completes and is returned on the root

    so that in the steady state the frames of the next step travel out and the packed frames of the previous step travel back
    while this step computes.  On a GPU the two exchanges are issued under their own HIP streams (RCCL orders its transfers
    behind the stream that is current at issue) and tied to the compute stream with events in both directions -- buffer reuse
    included; the host never blocks.  With host tensors (gloo tests) the same schedule runs with blocking waits.
    compute(frames_block, out_block): this rank's step, e.g. lambda f, o: engine.pipeline(f, p, sp, out=o).
    Results are identical to scatter_frames -> compute -> gather_outputs step by step (tests/test_shard_gloo.py)."""

    def __init__(self, n_frames: int, frame_shape: Tuple[int, int, int], out_shape: Tuple[int, int, int], device: torch.device,
                 compute: Callable, src: int = 0, out_dtype=torch.uint8):
        self.n, self.src, self.dev, self.compute = n_frames, src, device, compute
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        lo, hi = frame_range(n_frames, self.world, self.rank)
        self.inb = [torch.empty((hi - lo,) + tuple(frame_shape), dtype=torch.uint8, device=device) for _ in range(2)]
        self.outb = [torch.empty((hi - lo,) + tuple(out_shape), dtype=out_dtype, device=device) for _ in range(2)]
        self.full = [torch.empty((n_frames,) + tuple(out_shape), dtype=out_dtype, device=device) for _ in range(2)] if self.rank == src else [None, None]
        self.gpu = device.type == "cuda"
        if self.gpu:
            self.s_in, self.s_out = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
        self.k = 0
        self.scat = [None, None]                 # per buffer: pending scatter handle (+ event on s_in)
        self.gath = [None, None]
        self.ev_in = [None, None]                # scatter into inb[b] complete (recorded on s_in)
        self.ev_cmp = [None, None]               # compute that read inb[b] / wrote outb[b] complete (recorded on the compute stream)
        self.ev_out = [None, None]               # gather that read outb[b] complete (recorded on s_out)

    # -- stage helpers ---------------------------------------------------------------------------------------------------
    def _start_scatter(self, b, frames):
        if self.gpu:
            cur = torch.cuda.current_stream(self.dev)
            self.s_in.wait_stream(cur)                                   # the root's frames were produced on the caller's stream
            if self.ev_cmp[b] is not None:
                self.s_in.wait_event(self.ev_cmp[b])                     # inb[b] was read by compute(k - 2)
            with torch.cuda.stream(self.s_in):
                _, h = scatter_frames(frames, self.n, self.inb[b].shape[1:], self.dev, self.src, out=self.inb[b], async_op=True)
                h.wait()                                                 # stream-ordered for RCCL: s_in is behind the transfers
                self.ev_in[b] = torch.cuda.Event()
                self.ev_in[b].record(self.s_in)
            self.scat[b] = h
        else:
            _, self.scat[b] = scatter_frames(frames, self.n, self.inb[b].shape[1:], self.dev, self.src, out=self.inb[b], async_op=True)

    def _run_compute(self, b):
        if self.gpu:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(self.ev_in[b])
            if self.ev_out[b] is not None:
                cur.wait_event(self.ev_out[b])                           # outb[b] was read by gather(k - 2)
            self.compute(self.inb[b], self.outb[b])
            self.ev_cmp[b] = torch.cuda.Event()
            self.ev_cmp[b].record(cur)
        else:
            self.scat[b].wait()
            if self.gath[b] is not None:
                self.gath[b].wait()
            self.compute(self.inb[b], self.outb[b])

    def _start_gather(self, b):
        if self.gpu:
            self.s_out.wait_event(self.ev_cmp[b])
            with torch.cuda.stream(self.s_out):
                _, h = gather_outputs(self.outb[b], self.n, self.src, full=self.full[b], async_op=True)
                h.wait()
                self.ev_out[b] = torch.cuda.Event()
                self.ev_out[b].record(self.s_out)
            self.gath[b] = h
        else:
            _, self.gath[b] = gather_outputs(self.outb[b], self.n, self.src, full=self.full[b], async_op=True)

    def _finish_gather(self, b):
        if self.gpu:
            torch.cuda.current_stream(self.dev).wait_event(self.ev_out[b])
        else:
            self.gath[b].wait()
            self.gath[b] = None
        return self.full[b]

    # -- the schedule -------------------------------------------------------------------------------------------------------
    def submit(self, frames: Optional[torch.Tensor]):
        """Feed step k (the root passes [n_frames,H,W,3], the others None).  Returns the packed frames of step k - 2 on the root
        (None while the pipeline fills, and on the other ranks).  The returned buffer is valid until the next call."""
        k = self.k
        done = None
        if k >= 2:
            done = self._finish_gather(k & 1)            # gather(k - 2) wrote full[k & 1]: hand it out before that buffer's next turn
        self._start_scatter(k & 1, frames)
        if k >= 1:
            self._run_compute((k - 1) & 1)
            self._start_gather((k - 1) & 1)
        self.k += 1
        return done if self.rank == self.src else None

    def flush(self):
        """Drain: returns the list of the (up to two) results still in flight, oldest first (root; [] elsewhere)."""
        outs = []
        k = self.k
        if k >= 2:
            outs.append(self._finish_gather(k & 1))      # gather(k - 2)
        if k >= 1:
            b = (k - 1) & 1
            self._run_compute(b)
            self._start_gather(b)
            outs.append(self._finish_gather(b))
        self.k = 0
        self.scat, self.gath = [None, None], [None, None]
        return [o.clone() for o in outs] if self.rank == self.src else []
