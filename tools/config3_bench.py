#!/usr/bin/env python
"""BASELINE configs[2] alone (ViT-L, 3840x2160 -> Full-TAB) per engine precision and batch: frames/s + the depth error against the
committed reference fixture.  A/B aid for the e4m3 paths (D2S_LIB=... selects a variant library).
    python tools/config3_bench.py [--prec bf16 fp8 fp8_mlp] [--batch 1 8]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from desktop2stereo_amd import ops, synth
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
from desktop2stereo_amd.weights import make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--prec", nargs="+", default=["bf16", "fp8", "fp8_mlp"])
ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
ap.add_argument("--model", default="vitl")
a = ap.parse_args()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda")
cfg, H, W = MODELS[a.model], 2160, 3840
w = make_weights(cfg, 0)
h, w_, _ = engine_shape(H, W, 518)
p = PipelineParams(depth_resolution=518, display_mode="Full-TAB")
sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, "Full-TAB", p.fill_16_9)
oh, ow = ops.sbs_shape(H, W, sp)
ref = None
if a.model == "vitl":
    z = np.load(os.path.join(REPO, "tests", "golden", "vitl_r518_4k.npz"))
    fr = json.load(open(os.path.join(REPO, "tests", "golden", "vitl_r518_4k.json")))["frames"][0]
    ref = (z["f0_post_depth"], synth.structured_frame(fr["h"], fr["w"], fr["seed"]))
pool = [torch.from_numpy(synth.noise_frame(H, W, 9000 + j)[None]).to(dev) for j in range(2)]
for prec in a.prec:
    e = ops.Engine(cfg, w, h, w_, max_batch=max(a.batch), precision=prec)
    if prec != "bf16":
        e.calibrate(ops.preprocess(torch.from_numpy(synth.structured_frame(H, W, 0)).to(dev), 518))
    row = {"prec": prec}
    for B in a.batch:
        frames = torch.cat([pool[j & 1] for j in range(B)])
        out = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=dev)
        n = 40 if B == 1 else 12
        for _ in range(4): e.pipeline(frames, p, sp, use_ema=False, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): e.pipeline(frames, p, sp, use_ema=False, out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row[f"b{B}_fps"] = round(n * B / dt, 1)
    if ref is not None:
        post = ops.post_process_depth(e(ops.preprocess(torch.from_numpy(ref[1]).to(dev), 518)), p).cpu().numpy()[0]
        dd = np.abs(post - ref[0])
        row.update(l1=round(float(dd.mean()), 5), max=round(float(dd.max()), 5))
    print(json.dumps(row), flush=True)
    e.close()
