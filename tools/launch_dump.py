"""Per-launch HIP-event timing of ONE d2s_pipeline call (tuning aid): prints every recorded launch in order.

    D2S_PROF_DUMP=1 python tools/launch_dump.py [--model vitb] [--batch 1] [--res 518] [--prec bf16]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desktop2stereo_amd import ops, synth                     # noqa: E402
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape   # noqa: E402
from desktop2stereo_amd.weights import make_weights           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vitb")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--res", type=int, default=518)
ap.add_argument("--prec", default="bf16")
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--vda", action="store_true")
ap.add_argument("--mode", default="Full-SBS")
a = ap.parse_args()
os.environ["D2S_PROF_DUMP"] = "1"
cfg = MODELS[a.model]
h, w, _ = engine_shape(a.height, a.width, a.res)
if a.vda:
    from desktop2stereo_amd.vda_weights import make_vda_weights
    eng = ops.Engine(cfg, make_vda_weights(cfg, 0), h, w, 1, a.prec, temporal=True)
else:
    eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, a.batch, a.prec)
if a.prec.startswith("fp8"):
    eng.calibrate(ops.preprocess(torch.from_numpy(synth.structured_frame(a.height, a.width, 0)).cuda(), a.res))
p = PipelineParams(depth_resolution=a.res, display_mode=a.mode)
sp = ops.sbs_params(0.064, 4.0, 0.0, a.mode, False)
frames = torch.from_numpy(np.stack([synth.noise_frame(a.height, a.width, i) for i in range(a.batch)])).cuda()
for _ in range(40 if a.vda else 5):
    eng.pipeline(frames, p, sp)
torch.cuda.synchronize()
eng.profile(True)
eng.pipeline(frames, p, sp)
torch.cuda.synchronize()
r = eng.profile_read()
tot = sum(v["ms"] for v in r.values())
print({k: round(v["ms"], 4) for k, v in r.items()}, "sum ms", round(tot, 4), file=sys.stderr)
