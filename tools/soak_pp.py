#!/usr/bin/env python
"""Race screen for the hand-scheduled kernels of round 2: everything here must be bit-identical from run to run.
  * gemm_pp_kernel (counted vmcnt, raw barriers, K-split tail): many shapes x repeats through d2s_gemm_probe and the batched engine;
  * stereo_warp_gather (inline-asm prefetch with counted vmcnt, wave-private LDS windows): all modes, batch 1..16, 1080p and 4K.
    python tools/soak_pp.py [--reps 100]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from desktop2stereo_amd import ops, synth
from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
from desktop2stereo_amd.weights import make_weights

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(1)
bad = 0
for prec in ("bf16", "fp8"):
    for (M, N, K) in [(24896, 768, 3072), (24896, 3072, 768), (24896, 2304, 768), (21006, 768, 768), (6224, 3072, 768), (12448, 768, 3072),
                      (700, 512, 256), (49792, 768, 768), (513, 1024, 512)]:
        A = torch.randn(M, K, device=dev) * 0.5; W = torch.randn(N, K, device=dev) * 0.5; b = torch.randn(N, device=dev)
        first = ops.gemm_probe(A, W, b, prec, 256256)
        n = sum(0 if torch.equal(ops.gemm_probe(A, W, b, prec, 256256), first) else 1 for _ in range(a.reps))
        print(f"gemm_pp {prec} {M}x{N}x{K}: {n} of {a.reps} repeats differ", flush=True); bad += n
        del A, W, first
cfg = MODELS["vitb"]; H, W_ = 1080, 1920
h, w, _ = engine_shape(H, W_, 518)
p = PipelineParams(depth_resolution=518)
for prec, B in (("bf16", 32), ("bf16", 27), ("fp8", 32)):
    eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, B, prec)
    frames = torch.from_numpy(np.stack([synth.structured_frame(H, W_, i) for i in range(B)])).to(dev)
    if prec == "fp8":
        eng.calibrate(ops.preprocess(frames[:2], 518))
    sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, "Full-SBS", p.fill_16_9)
    first = eng.pipeline(frames, p, sp).clone()
    n = sum(0 if torch.equal(eng.pipeline(frames, p, sp), first) else 1 for _ in range(max(10, a.reps // 4)))
    print(f"engine {prec} batch {B} (K-split tail, ping-pong linears, lane-strided warp): {n} repeats differ", flush=True); bad += n
    eng.close(); del frames, first
for (Hh, Ww, B) in ((1080, 1920, 1), (1080, 1920, 16), (2160, 3840, 3), (720, 1280, 5)):
    img = torch.from_numpy(np.stack([synth.noise_frame(Hh, Ww, i) for i in range(B)])).to(dev)
    dep = torch.from_numpy(np.stack([synth.smooth_depth(294, 518, i) for i in range(B)])).to(dev)
    for mode in ("Full-SBS", "Full-TAB", "Half-TAB", "Half-SBS"):
        for ratio in (4.0, 40.0):
            sp = ops.sbs_params(0.064, ratio, 0.05, mode, True)
            first = ops.make_sbs(img, dep, sp).clone()
            n = sum(0 if torch.equal(ops.make_sbs(img, dep, sp), first) else 1 for _ in range(a.reps))
            if n: print(f"warp {mode} {Ww}x{Hh} B={B} ratio {ratio}: {n} of {a.reps} repeats differ", flush=True)
            bad += n
    print(f"warp {Ww}x{Hh} B={B}: done", flush=True)
print("RACE SCREEN:", "FAILED" if bad else "clean")
