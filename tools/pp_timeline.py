#!/usr/bin/env python
"""In-kernel timeline of the ping-pong GEMM inside the engine (needs a library built with D2S_HIPCC_DEFS=-DD2S_PP_TIMING):
per epilogue kind, the last launch's per-block stamps -> main-loop / epilogue / operand-wait durations per tile.
    D2S_HIPCC_DEFS=-DD2S_PP_TIMING python -m desktop2stereo_amd.build --force;  python tools/pp_timeline.py --batch 32"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from desktop2stereo_amd import _lib, ops, synth
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
from desktop2stereo_amd.weights import make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--blocks", type=int, nargs="*", default=[0, 1, 8, 100, 255])
ap.add_argument("--kinds", type=int, nargs="*", default=None, help="epilogue kinds (gemm_pp.hip PP_K_*); default: those the engine uses at this batch")
a = ap.parse_args()
lib = _lib.load()
if not hasattr(lib, "d2s_pp_timing"):
    raise SystemExit("library was built without -DD2S_PP_TIMING")
lib.d2s_pp_timing.argtypes = [C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
cfg = MODELS["vitb"]
H, W, B = 1080, 1920, a.batch
p = PipelineParams(depth_resolution=518, display_mode="Full-SBS")
h, w, _ = engine_shape(H, W, 518)
eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, max_batch=B, precision="bf16")
sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, "Full-SBS", p.fill_16_9)
oh, ow = ops.sbs_shape(H, W, sp)
frames = torch.from_numpy(np.stack([synth.noise_frame(H, W, i) for i in range(B)])).to(dev)
out = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=dev)
for _ in range(3):
    eng.pipeline(frames, p, sp, use_ema=False, out=out)
torch.cuda.synchronize()
names = {0: "bf16 (proj-like / plain)", 1: "FC1 (GELU, bf16)", 2: "QKV", 3: "f32 residual (proj, FC2)",
         4: "FC1, LayerNorm folded (consumer)", 5: "QKV, LayerNorm folded (consumer)", 6: "f32 residual + LN statistics (proj, FC2: producer)"}
folded = os.environ.get("D2S_LNF_PP", "1") != "0"
for kind in (a.kinds if a.kinds is not None else ((5, 4, 6) if folded else (2, 1, 3))):
    assert lib.d2s_pp_timing(kind, None) == 0
    eng.pipeline(frames, p, sp, use_ema=False, out=out)
    torch.cuda.synchronize()
    buf = np.zeros(264 * 64, dtype=np.uint64)
    assert lib.d2s_pp_timing(-2, buf.ctypes.data_as(C.c_void_p)) == 0
    tw = buf.reshape(264, 64).astype(np.float64)[256:] * 0.01     # per-wave stamps of block 8
    t = buf.reshape(264, 64).astype(np.float64)[:256] * 0.01      # us (100 MHz)
    live = t[:, 0] > 0
    t0 = t[live, 0].min()
    print(f"== kind {kind}: {names[kind]}  (last such launch of the frame; {int(live.sum())} blocks)")
    print(f"   block start spread: {t[live, 0].max() - t0:6.2f} us")
    ntile = int(((t[live, 1:] > 0).sum(axis=1).max() + 3) // 4)
    for k in range(ntile):
        s = t[:, 1 + 4 * k: 5 + 4 * k]
        has = live & (s[:, 2] > 0)
        prev = t[:, 0] if k == 0 else t[:, 4 * k]
        def st(x):
            return f"{x.mean():6.2f} [{x.min():6.2f} {x.max():6.2f}]"
        line = f"   tile {k}: {int(has.sum()):3d} blocks  lead-in {st(s[has, 0] - prev[has])}  main {st(s[has, 1] - s[has, 0])}  epilogue {st(s[has, 2] - s[has, 1])}"
        nxt = has & (s[:, 3] > 0)
        if nxt.any():
            line += f"  next-operand wait {st(s[nxt, 3] - s[nxt, 2])}"
        print(line)
    end = np.where(t[:, 1:] > 0, t[:, 1:], 0).max(axis=1)
    print(f"   block end: mean {end[live].mean() - t0:7.2f}  max {end[live].max() - t0:7.2f} us after the first block started")
    if live[8]:                                    # per-wave stamps of block 8
        print("   block 8, per wave: main-loop end / epilogue end relative to wave 0's main-loop end, per tile")
        for k in range(ntile):
            base = t[8, 2 + 4 * k]
            if base <= 0: break
            print(f"     tile {k}: " + "  ".join(f"w{w}: {tw[w, 2 + 4 * k] - base:+5.2f} / {tw[w, 3 + 4 * k] - base:+5.2f}" for w in range(8)))
    for b in a.blocks:
        if b < 256 and live[b]:
            print(f"   block {b:3d}: " + " ".join(f"{x - t0:6.1f}" for x in t[b] if x > 0))
lib.d2s_pp_timing(-1, None)
