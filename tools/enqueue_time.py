#!/usr/bin/env python
"""Host time to ENQUEUE one batch-1 frame (d2s_pipeline returns when its ~99 launches are queued) against the frame's GPU time:
is the launch chain ever waiting for the host?      python tools/enqueue_time.py"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from desktop2stereo_amd import ops, synth
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
from desktop2stereo_amd.weights import make_weights

H, W, res, mode = 1080, 1920, 518, "Full-SBS"
cfg = MODELS["vitb"]; h, w, _ = engine_shape(H, W, res)
eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, 1, "bf16")
p = PipelineParams(depth_resolution=res, display_mode=mode); sp = ops.sbs_params(0.064, 4.0, 0.0, mode, False)
frames = torch.from_numpy(np.stack([synth.noise_frame(H, W, 0)])).cuda()
oh, ow = ops.sbs_shape(H, W, sp); out = torch.empty((1, oh, ow, 3), dtype=torch.uint8, device="cuda")
for _ in range(50): eng.pipeline(frames, p, sp, use_ema=False, out=out)
torch.cuda.synchronize()
for n in (1, 8, 200):
    t0 = time.perf_counter()
    for _ in range(n): eng.pipeline(frames, p, sp, use_ema=False, out=out)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{n:4d} frames: enqueue {1e3 * (t1 - t0) / n:.3f} ms per frame, until the GPU is done {1e3 * (t2 - t0) / n:.3f} ms per frame", flush=True)
