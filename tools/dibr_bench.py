#!/usr/bin/env python
"""d2s_dibr_warp alone: 1080p scene with hard depth edges -> both eyes; us per launch and HBM fraction.
    python tools/dibr_bench.py [--mode Full-SBS] [--batch 1 8]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desktop2stereo_amd import ops, synth
ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="Full-SBS")
ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
ap.add_argument("--height", type=int, default=1080); ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--kind", default="boxes")
a = ap.parse_args()
dev = torch.device("cuda")
img, dep = synth.dibr_scene(a.height, a.width, 11, a.kind)
f1, d1 = torch.from_numpy(img).to(dev)[None], torch.from_numpy(dep).to(dev)[None]
dp = ops.dibr_params(display_mode=a.mode)
for B in a.batch:
    f, d = f1.expand(B, -1, -1, -1).contiguous(), d1.expand(B, -1, -1).contiguous()
    for _ in range(3): out = ops.dibr_warp(f, d, dp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.dibr_warp(f, d, dp)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nbytes = B * (a.height * a.width * 7) + out.numel()
    print(f"{a.mode} {a.kind} B={B}: {us:8.1f} us / launch  {us / B:7.1f} us / frame  {nbytes / us / 1e3:7.1f} GB/s = {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
