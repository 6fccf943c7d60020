"""In-kernel tail reduce of the ping-pong GEMM (gemm_pp.hip, D2S_PP_INK) against the two-launch path: the engine output must be
bit-identical (same slab order), run after run.     python tools/ink_check.py [batch ...]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from desktop2stereo_amd import ops
from desktop2stereo_amd.config import MODELS, engine_shape
from desktop2stereo_amd.weights import make_weights

batches = [int(a) for a in sys.argv[1:]] or [32, 27, 16, 12]
cfg = MODELS["vitb"]
wts = make_weights(cfg, 0)
h, w, _ = engine_shape(1080, 1920, 518)
bad = 0
for B in batches:
    x = torch.randn(B, 3, h, w, device="cuda", generator=torch.Generator(device="cuda").manual_seed(B))
    outs = {}
    for ink in ("0", "1", "2"):
        os.environ["D2S_PP_INK"] = ink
        ops.reload_env()
        eng = ops.Engine(cfg, wts, h, w, B, "bf16")
        o = eng(x).cpu().numpy()
        rep = 0
        for _ in range(30):
            rep += int(not np.array_equal(o, eng(x).cpu().numpy()))
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            eng(x)
        t1.record(); torch.cuda.synchronize()
        outs[ink] = o
        print(f"batch {B} D2S_PP_INK={ink}: {t0.elapsed_time(t1) / 10:.3f} ms per forward, run-to-run differences in 30 repeats: {rep}, finite: {np.isfinite(o).all()}")
        bad += rep
        eng.close()
    same = np.array_equal(outs["0"], outs["1"]) and np.array_equal(outs["0"], outs["2"])
    print(f"batch {B}: in-kernel == two-launch: {same}")
    bad += int(not same)
sys.exit(1 if bad else 0)
