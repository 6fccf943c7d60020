#!/usr/bin/env python
"""One engine at batch 2n on one stream against two engines at batch n on two streams (independent frames: the second form lets the
bandwidth-bound launches of one half-batch -- proj / FC2 epilogues, warp -- run under the matrix-bound launches of the other).
    python tools/two_streams_bench.py [--batch 32] [--steps 40]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from desktop2stereo_amd import ops, synth
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
from desktop2stereo_amd.weights import make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--model", default="vitb")
a = ap.parse_args()
H, W, res, mode = 1080, 1920, 518, "Full-SBS"
cfg = MODELS[a.model]; h, w, _ = engine_shape(H, W, res)
wts = make_weights(cfg, 0)
p = PipelineParams(depth_resolution=res, display_mode=mode); sp = ops.sbs_params(0.064, 4.0, 0.0, mode, False)
B, n = a.batch, a.batch // 2
frames = torch.from_numpy(np.stack([synth.noise_frame(H, W, i) for i in range(B)])).cuda()
oh, ow = ops.sbs_shape(H, W, sp)


def timed(fn, warm, steps):
    for i in range(warm): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): fn(i)
    torch.cuda.synchronize(); return time.perf_counter() - t0


eng = ops.Engine(cfg, wts, h, w, B, "bf16")
out = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device="cuda")
dt1 = timed(lambda i: eng.pipeline(frames, p, sp, use_ema=False, out=out), 5, a.steps)
print(f"one engine, batch {B}, one stream: {a.steps * B / dt1:8.1f} frames/s ({1e3 * dt1 / a.steps:.3f} ms per step)", flush=True)
eng.close()
engs = [ops.Engine(cfg, wts, h, w, n, "bf16") for _ in range(2)]
outs = [torch.empty((n, oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
halves = [frames[:n].contiguous(), frames[n:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()


def two(i):
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            engs[k].pipeline(halves[k], p, sp, use_ema=False, out=outs[k])


dt2 = timed(two, 5, a.steps)
print(f"two engines, batch {n} each, two streams: {a.steps * B / dt2:8.1f} frames/s ({1e3 * dt2 / a.steps:.3f} ms per pair)", flush=True)
ok = torch.equal(torch.cat(outs), out)
print("outputs identical to the single-engine batch:", ok)
