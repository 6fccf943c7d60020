#!/usr/bin/env python
"""Micro-benchmark of the stereo-warp kernel (d2s_make_sbs, u8 HWC in/out, depth at model res)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from desktop2stereo_amd import ops, synth, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--hw", type=int, nargs=2, default=[1080, 1920])
ap.add_argument("--modes", nargs="+", default=["Full-SBS", "Half-SBS", "Full-TAB", "Half-TAB"]); ap.add_argument("--no-dibr", action="store_true")
ap.add_argument("--digest", action="store_true", help="print a sha256 of every mode's output (A/B builds must agree: D2S_LIB=...)")
a = ap.parse_args()
dev = torch.device("cuda"); H, W = a.hw; B = a.batch
img = torch.from_numpy(np.stack([synth.noise_frame(H, W, i) for i in range(B)])).to(dev)
dep = torch.from_numpy(np.stack([synth.smooth_depth(294, 518, i) for i in range(B)])).to(dev)
for mode in a.modes:
    sp = ops.sbs_params(0.064, 4.0, 0.0, mode, True)
    oh, ow = ops.sbs_shape(H, W, sp)
    for _ in range(20): ops.make_sbs(img, dep, sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200; e0.record()
    for _ in range(n): ops.make_sbs(img, dep, sp)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    byts = B * (H * W * 3 + 294 * 518 * 4 + oh * ow * 3)
    print(f"{mode:9s} B={B} {W}x{H}: {us:8.1f} us  {byts/us/1e6:7.3f} TB/s algorithmic ({byts/1e6:.2f} MB)  [incl. output alloc]", flush=True)
    if a.digest:
        import hashlib
        for conv, sub in ((0.0, "conv 0"), (0.6, "conv 0.6, wide shifts")):
            o = ops.make_sbs(img, dep * (1.0 if conv == 0.0 else 3.0), ops.sbs_params(0.064 if conv == 0.0 else 0.3, 4.0, conv, mode, True))
            print(f"  sha256[{sub}] {hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
if a.no_dibr: sys.exit(0)
# the viewer-shader warp with disocclusion in-painting (d2s_dibr_warp): full-resolution depth in, both eyes out
depf = torch.from_numpy(np.stack([synth.smooth_depth(H, W, i) for i in range(B)])).to(dev)
for mode in ("Full-SBS", "Half-SBS"):
    dp = ops.dibr_params(0.064, 4.0, 0.0, mode)
    for _ in range(5): ops.dibr_warp(img, depf, dp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50; e0.record()
    for _ in range(n): out = ops.dibr_warp(img, depf, dp)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    byts = B * (H * W * 3 + H * W * 4) + out.numel()
    print(f"DIBR {mode:9s} B={B} {W}x{H}: {us:8.1f} us  {byts/us/1e6:7.3f} TB/s algorithmic ({byts/1e6:.2f} MB)  [incl. output alloc]", flush=True)
