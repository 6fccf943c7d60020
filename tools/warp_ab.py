#!/usr/bin/env python
"""A/B of the stereo-warp kernels on the GPU: the fast path (stereo_warp_gather, D2S_WARP_GATHER=1) against D2S_WARP_GATHER=0 -- the
generic per-pixel float kernel since round 6's clean-up (until then: round 5's LDS-staged kernels, the "staged" column of
profiles/r6_01) -- time per launch (HIP events, output pre-allocated) and byte differences, per display mode and batch."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from desktop2stereo_amd import ops, synth, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, nargs="+", default=[1, 32]); ap.add_argument("--hw", type=int, nargs=2, default=[1080, 1920])
ap.add_argument("--modes", nargs="+", default=["Full-SBS", "Half-SBS", "Full-TAB", "Half-TAB"]); ap.add_argument("--ratio", type=float, default=4.0)
ap.add_argument("--kind", default="noise"); ap.add_argument("--n", type=int, default=100)
ap.add_argument("--full-depth", action="store_true", help="depth map of the frame's size (the drop-in make_sbs surface) instead of 294 x 518")
a = ap.parse_args()
dev = torch.device("cuda"); H, W = a.hw
lib = _lib.load()
def setenv(k, v):
    os.environ[k] = str(v); lib.d2s_debug_reload_env()
for B in a.batches:
    gen = synth.noise_frame if a.kind == "noise" else synth.structured_frame
    img = torch.from_numpy(np.stack([gen(H, W, i) for i in range(B)])).to(dev)
    dep = torch.from_numpy(np.stack([synth.smooth_depth(*((H, W) if a.full_depth else (294, 518)), i) for i in range(B)])).to(dev)
    for mode in a.modes:
        sp = ops.sbs_params(0.064, a.ratio, 0.0, mode, True)
        oh, ow = ops.sbs_shape(H, W, sp)
        outs, times = {}, {}
        for gather in (0, 1):
            setenv("D2S_WARP_GATHER", gather)
            out = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=dev)
            for _ in range(10): ops.make_sbs(img, dep, sp, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.n): ops.make_sbs(img, dep, sp, out=out)
            e1.record(); torch.cuda.synchronize()
            times[gather] = e0.elapsed_time(e1) / a.n * 1e3
            outs[gather] = out.cpu().numpy().astype(np.int16)
        d = np.abs(outs[1] - outs[0])
        byts = B * (H * W * 3 + dep.shape[-2] * dep.shape[-1] * 4 + oh * ow * 3)
        print(f"{mode:9s} B={B:2d} {W}x{H}: other {times[0]:7.1f} us ({byts/times[0]/1e6:5.2f} TB/s)  gather {times[1]:7.1f} us ({byts/times[1]/1e6:5.2f} TB/s = {byts/times[1]/8e6:.3f} of 8 TB/s)"
              f"  | bytes differing: {(d > 0).mean():.2e}, max {d.max()} LSB", flush=True)
