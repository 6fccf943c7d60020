#!/usr/bin/env python
"""Micro-benchmark + bit-identity check of the depth post-process (d2s_post_process_to) at model resolution.

    python tools/post_bench.py [--batches 1 2 4] [--hw 294 518]

For every batch: D2S_POST_ONE=1 (post_fused_kernel: bounds + shape + both blurs in one launch) against D2S_POST_ONE=0 (round 4:
percentile_bounds_kernel + shape_blur_kernel / shape_hblur + vblur), outputs compared bit for bit, µs per call from HIP events."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np      # noqa: E402
import torch            # noqa: E402

from desktop2stereo_amd import ops, synth                 # noqa: E402
from desktop2stereo_amd.config import PipelineParams      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4])
ap.add_argument("--hw", type=int, nargs=2, default=[294, 518])
ap.add_argument("--no-check", action="store_true", help="timing-only library variants (PF_CUT builds): skip the bit-identity assertion")
a = ap.parse_args()
dev = torch.device("cuda")
h, w = a.hw
p = PipelineParams()


def run(x, n=300):
    for _ in range(20):
        out = ops.post_process_depth_to(x, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = ops.post_process_depth_to(x, p)
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / n * 1e3


for B in a.batches:
    rng = np.random.default_rng(B)
    x = np.stack([synth.smooth_depth(h, w, i) * 20.0 + rng.random((h, w), dtype=np.float32) * 0.3 for i in range(B)]).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    res = {}
    for one in ("0", "1"):
        os.environ["D2S_POST_ONE"] = one
        ops.reload_env()
        res[one] = run(xt)
    same = bool(torch.equal(res["0"][0], res["1"][0]))
    print(f"post-process B={B} {w}x{h}: separate launches {res['0'][1]:7.1f} us | one launch {res['1'][1]:7.1f} us | bit-identical: {same}"
          f"  [incl. output alloc]", flush=True)
    assert same or a.no_check
os.environ.pop("D2S_POST_ONE", None)
ops.reload_env()
