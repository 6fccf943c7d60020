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


def ema_step_torch(depth, state, initialised, alpha):
    """Test stand-in for the HIP d2s_ema_update on host tensors (the package itself computes on the GPU only): one
    DepthStabilizer step, reference depth.py:1873-1887 -- first frame seeds the state and passes through, later frames return
    prev.lerp_(depth, 1 - alpha); `depth` is overwritten with the returned map."""
    if not initialised:
        state.copy_(depth)
        return depth
    state.lerp_(depth, 1.0 - alpha)
    depth.copy_(state)
    return depth


# ---- EMA under frame sharding: gather to the stream owner -> scan -> send back ------------------------
def _ema_worker(rank, world, port, owner, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from desktop2stereo_amd.shard import ema_exchange, frame_range
    z = np.load(os.path.join(REPO, "tests", "golden", "tiny_r84.npz"))
    n = 3
    lo, hi = frame_range(n, world, rank)
    mine = torch.from_numpy(np.stack([z[f"f{i}_post_depth"] for i in range(lo, hi)]) if hi > lo else np.zeros((0,) + z["f0_post_depth"].shape, np.float32))
    state = torch.zeros(z["f0_post_depth"].shape, dtype=torch.float32)
    out, init = ema_exchange(mine.clone(), n, state, False, 0.9, owner=owner, ema_step=ema_step_torch)
    # every frame's stabilised map == the REFERENCE's single-stream EMA chain (DepthStabilizer, alpha 0.9; golden tiny_r84)
    errs = [float(np.abs(out[i - lo].numpy() - z[f"f{i}_ema_state"]).max()) for i in range(lo, hi)]
    if rank == owner:
        errs.append(float(np.abs(state.numpy() - z["f2_ema_state"]).max()))      # the owner keeps the stream's state
    q.put((rank, errs, bool(init)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,owner", [(2, 0), (2, 1), (3, 1)])
def test_ema_exchange_matches_single_rank_chain(world, owner):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + 7 * world + owner) % 2000
    procs = [ctx.Process(target=_ema_worker, args=(r, world, port, owner, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, errs, init in got:
        assert init and all(e <= 2e-6 for e in errs), (rank, errs)
    assert sum(len(e) for _, e, _ in got) == 3 + 1


def test_ema_exchange_has_no_cpu_arithmetic():
    """Host tensors without an explicit step: refused (the product computes on the GPU only)."""
    from desktop2stereo_amd.shard import ema_exchange
    with pytest.raises(RuntimeError):
        ema_exchange(torch.zeros(1, 2, 2), 1, torch.zeros(2, 2), False, 0.9)


# ---- rank-0 ingest as a software pipeline: scatter(k + 1) | compute(k) | gather(k - 1) ---------------------------------------
def _pipe_worker(rank, world, port, n_frames, steps, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from desktop2stereo_amd.shard import PipelinedIngest, gather_outputs, scatter_frames
    H, W = 5, 7
    dev = torch.device("cpu")
    rng = np.random.default_rng(3)
    seq = [torch.from_numpy(rng.integers(0, 256, (n_frames, H, W, 3), dtype=np.uint8)) for _ in range(steps)]       # same on every rank (seeded)
    calls = []

    def compute(f, o):                                   # stand-in for engine.pipeline: a per-frame map that depends on the call index
        calls.append(f.shape[0])
        o.copy_(torch.cat([f, (f.to(torch.int32) * 3 % 251).to(torch.uint8)], dim=2))
    # reference schedule: step by step, nothing overlapped
    want = []
    for k in range(steps):
        mine = scatter_frames(seq[k] if rank == 0 else None, n_frames, (H, W, 3), dev)
        o = torch.empty((mine.shape[0], H, 2 * W, 3), dtype=torch.uint8)
        compute(mine, o)
        g = gather_outputs(o, n_frames)
        if rank == 0:
            want.append(g.clone())
    # pipelined schedule
    pipe = PipelinedIngest(n_frames, (H, W, 3), (H, 2 * W, 3), dev, compute)
    got = []
    for k in range(steps):
        r = pipe.submit(seq[k] if rank == 0 else None)
        if rank == 0:
            assert (r is None) == (k < 2), k
            if r is not None:
                got.append(r.clone())
        else:
            assert r is None
    got += pipe.flush()
    dist.barrier()
    if rank == 0:
        q.put(len(got) == steps and all(torch.equal(a, b) for a, b in zip(got, want))
              and torch.equal(want[0], torch.cat([seq[0], (seq[0].to(torch.int32) * 3 % 251).to(torch.uint8)], dim=2)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,steps", [(2, 5, 6), (3, 7, 5), (2, 1, 3), (3, 2, 1)])
def test_pipelined_ingest_equals_stepwise(world, n_frames, steps):
    """Results of the overlapped schedule == scatter -> compute -> gather step by step, every step, incl. the drain, ragged
    blocks, ranks with no frames and runs shorter than the pipeline depth."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() + 11 * world + n_frames + steps) % 2000
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, n_frames, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=180)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ok is True


# ---- one stream, frames sharded, temporal smoothing on: ShardedStream == the single-rank chain -------------------------------
class _GoldenStages:
    """Stand-in stages on host tensors: depth_small returns the reference's post-processed maps of the golden frames (looked up by
    the frame's tag pixel), warp returns the depth it was given -- what reaches the warp is exactly the stabilised map."""

    def __init__(self, z):
        self.z = z

    def depth_small(self, frames):
        return torch.from_numpy(np.stack([self.z[f"f{int(f[0, 0, 0])}_post_depth"] for f in frames]).astype(np.float32)) if frames.shape[0] else \
            torch.zeros((0,) + self.z["f0_post_depth"].shape, dtype=torch.float32)

    def warp(self, frames, depth_small):
        return depth_small


def _stream_worker(rank, world, port, stream_id, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from desktop2stereo_amd.shard import ShardedStream, frame_range, stream_owner
    z = np.load(os.path.join(REPO, "tests", "golden", "tiny_r84.npz"))
    st = ShardedStream(_GoldenStages(z), stream_id=stream_id, alpha=0.9, ema_step=ema_step_torch)
    assert st.owner == stream_owner(stream_id, world)
    errs = []
    # two calls on the same stream: frames 0-1, then frame 2 -- the state carries over on the owner (the golden chain is 3 frames)
    for first, n in ((0, 2), (2, 1)):
        lo, hi = frame_range(n, world, rank)
        frames = torch.zeros((hi - lo, 2, 2, 3), dtype=torch.uint8)
        for i in range(hi - lo):
            frames[i, 0, 0, 0] = first + lo + i                    # tag: which golden frame this is
        out = st(frames, n)
        errs += [float(np.abs(out[i].numpy() - z[f"f{first + lo + i}_ema_state"]).max()) for i in range(hi - lo)]
    q.put((rank, errs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,stream_id", [(2, 0), (2, 1), (3, 2)])
def test_sharded_stream_matches_reference_chain(world, stream_id):
    """predict_depth(use_temporal_smooth=True) for a stream whose frames are sharded: every frame's stabilised map == the
    REFERENCE's single-stream DepthStabilizer chain (golden tiny_r84), across two calls."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + (os.getpid() + 13 * world + stream_id) % 2000
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, stream_id, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sum(len(e) for _, e in got) == 3
    for rank, errs in got:
        assert all(e <= 2e-6 for e in errs), (rank, errs)


# ---- bench.py's rank body under gloo, world_size 2 -------------------------------------------------------------------
class _StandInEngine:
    """CPU stand-in for ops.Engine, injected into bench.rank_body by this test only: a deterministic per-frame map with the
    engine's call shape (uint8 [B,H,W,3] -> Full-SBS uint8 [B,H,2W,3]), so the RANK logic of bench.py -- WORLD_SIZE handling,
    rank counting by all-reduce, own-frames and rank-0-ingest modes (scatter -> step -> gather), barriers, the max-over-ranks
    clock, one JSON object from rank 0 -- runs without a GPU.  It computes nothing of the product."""

    def __init__(self, max_batch):
        self.max_batch = max_batch
        self.calls = 0

    @staticmethod
    def sbs_params(ipd, ratio, conv, mode, fill):
        assert mode == "Full-SBS"
        return mode

    @staticmethod
    def sbs_shape(H, W, sp):
        return H, 2 * W

    def pipeline(self, frames, p, sp, use_ema=False, out=None):
        assert frames.dtype == torch.uint8 and frames.shape[0] <= self.max_batch
        self.calls += 1
        out.copy_(torch.cat([frames, 255 - frames], dim=2))
        return out

    def close(self):
        pass


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      D2S_DIST_BACKEND="gloo")
    import bench
    args = bench.parse_args(["--gpus", str(world), "--steps", "6", "--warmup", "2", "--batch", "3", "--height", "6", "--width", "8",
                             "--no-profile", "--no-cpu-baseline", "--sink-quality", "0", "--also-batch", "0"])
    res = bench.rank_body(args, engine_factory=_StandInEngine, device=torch.device("cpu"))
    if rank == 0:
        q.put(res)
    else:
        assert res is None


def test_bench_rank_body_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["scaling"] == "weak"
    assert res["steps"] == 6 and res["warmup"] == 2 and res["value"] > 0
    assert abs(res["value"] - 6 * 3 * 2 / (res["ms_per_step"] * 6e-3)) < 1e-6 * res["value"]       # whole-job frames / max-rank time
    ing = res["ingest_rank0"]
    assert ing["frames_per_step"] == 6 and ing["value"] > 0 and "isend" in ing["exchange"]
    json_line = __import__("json").dumps(res)
    assert "\n" not in json_line


def test_bench_refuses_mismatched_world(monkeypatch):
    """--gpus N must mean N ranks: with WORLD_SIZE set to something else bench.py stops instead of reporting N."""
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.setenv("WORLD_SIZE", "1")
    args = bench.parse_args(["--gpus", "4"])
    with pytest.raises(SystemExit):
        bench.rank_body(args, engine_factory=_StandInEngine, device=torch.device("cpu"))
