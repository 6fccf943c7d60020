#!/usr/bin/env python
"""Experiment: L independent engines ("lanes") on L HIP streams, B/L frames each, against one engine at batch B.
Does the hardware scheduler fill the tile-quantisation tails of one lane's launches with the other lane's blocks?
    python tools/lanes_exp.py --batch 32 --lanes 2"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from desktop2stereo_amd import ops, synth
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
from desktop2stereo_amd.weights import make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[32, 16, 2, 1])
ap.add_argument("--lanes", type=int, nargs="+", default=[1, 2])
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = MODELS["vitb"]
H, W = 1080, 1920
p = PipelineParams(depth_resolution=518, display_mode="Full-SBS")
h, w, _ = engine_shape(H, W, 518)
weights = make_weights(cfg, 0)
sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, "Full-SBS", p.fill_16_9)
oh, ow = ops.sbs_shape(H, W, sp)

for B in a.batch:
    for L in a.lanes:
        if L > 1 and B == 1:
            nb, Bt = 1, L                       # L frames in flight, one per lane
        elif B % L:
            continue
        else:
            nb, Bt = B // L, B
        engs = [ops.Engine(cfg, weights, h, w, max_batch=nb, precision="bf16") for _ in range(L)]
        streams = [torch.cuda.Stream() for _ in range(L)]
        frames = [torch.from_numpy(np.stack([synth.noise_frame(H, W, 7 * l + i) for i in range(nb)])).to(dev) for l in range(L)]
        outs = [torch.empty((nb, oh, ow, 3), dtype=torch.uint8, device=dev) for _ in range(L)]

        def step():
            for l in range(L):
                with torch.cuda.stream(streams[l]):
                    engs[l].pipeline(frames[l], p, sp, use_ema=False, out=outs[l])
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"frames/step {Bt:3d}  lanes {L}  ({nb}/lane): {1e3 * dt / a.steps:8.3f} ms/step  {a.steps * Bt / dt:8.1f} fps", flush=True)
        for e in engs:
            e.close()
        del engs
