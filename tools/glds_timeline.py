"""Per-block phase stamps of gemm_glds_kernel (build with D2S_HIPCC_DEFS=-DD2S_GLDS_TIMING; `python -m desktop2stereo_amd.build --force`).
Runs the ViT-B batch-1 encoder linears through d2s_gemm_probe and prints, per launch: when blocks start (dispatch ramp), how long
a block spends priming its ring / waiting for the first K tile / in the K loop / in the epilogue, and the launch's span.
    python tools/glds_timeline.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from desktop2stereo_amd import _lib, ops

lib = C.CDLL(_lib.LIB_PATH)
lib.d2s_glds_timing.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
SHAPES = {"QKV": (778, 2304, 768), "proj": (778, 768, 768), "FC1": (778, 3072, 768), "FC2": (778, 768, 3072)}
if "--neck32" in sys.argv:         # the batch-32 shapes of the DPT neck's small-K linears (ConvT k4 / k2, reassemble 1x1, fusion 1x1, patch embed)
    SHAPES = {"convT4": (24864, 1536, 96), "convT2": (24864, 768, 192), "reasm96": (24864, 96, 768), "fusion1x1": (397824, 128, 128), "patch": (24864, 768, 640)}
COLD = "--cold" in sys.argv        # run the OTHER shapes' kernels (different instantiations: ~250 KB of code) right before the measured launch
ops_in = {n: (torch.randn(M, K, device=dev) * 0.5, torch.randn(N, K, device=dev) * 0.5, torch.randn(N, device=dev)) for n, (M, N, K) in SHAPES.items()}
if COLD:
    print("instruction cache cold: the other three shapes run between the warm-up and the measured launch")
for name, (M, N, K) in SHAPES.items():
    A, W, b = ops_in[name]
    for _ in range(3):
        ops.gemm_probe(A, W, b, "bf16", 0)
    if COLD:
        for other, (A2, W2, b2) in ops_in.items():
            if other != name:
                ops.gemm_probe(A2, W2, b2, "bf16", 0)
    torch.cuda.synchronize()
    lib.d2s_glds_timing(None, 1)
    ops.gemm_probe(A, W, b, "bf16", 0)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (4096 * 8))()
    lib.d2s_glds_timing(buf, 0)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    live = t[:, 4] > 0
    if not live.any():
        print(f"{name} {M}x{N}x{K}: no stamps (slots set: {[(int((t[:, k] > 0).sum())) for k in range(5)]}) -- the launch took another kernel")
        continue
    t = t[live]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0                                   # 100 MHz -> microseconds
    d = lambda a, b_: us[:, b_] - us[:, a]
    q = lambda x: f"{np.percentile(x, 10):5.1f} / {np.median(x):5.1f} / {np.percentile(x, 90):5.1f} / {x.max():5.1f}"
    print(f"{name} {M}x{N}x{K}: {len(t)} blocks, launch span {us[:, 4].max():.1f} us (first block start -> last block end)")
    print(f"   block start (p10 / median / p90 / max)   {q(us[:, 0])}")
    print(f"   entry -> ring primed                      {q(d(0, 1))}")
    print(f"   primed -> first K tile landed             {q(d(1, 2))}")
    print(f"   K loop                                    {q(d(2, 3))}")
    print(f"   epilogue (incl. store drain)              {q(d(3, 4))}")
    print(f"   block lifetime                            {q(d(0, 4))}")
