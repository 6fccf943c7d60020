#!/usr/bin/env python
"""VERDICT r5 item 3, priced before it is built: the batch-1 encoder linears (M = 778) on LARGER tiles x split-K.
For every (shape, tile, K ranges) the GEMM kernel + splitk_reduce_kernel pair is timed back to back (d2s_gemm_probe, D2S_SPLITK_FORCE);
an in-kernel reduce would replace the second launch (6.1 us stand-alone) by an exchange of >= 2-3 us (MI355X_MICROARCH.md price list:
handoff-flag 1.7-2.9 us under load), so `pair - 6.1 + 2.5` is the optimistic estimate of the fused form.
    D2S_GEMM_DEEP=0 python tools/splitk_probe.py          # sweep (general instantiations: the lean rings take one K range per block)
    python tools/splitk_probe.py --baseline               # what the engine runs today (tile rule, lean deep rings)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desktop2stereo_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--baseline", action="store_true"); ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda")
def t(A, W, tile, n):
    ops.gemm_probe(A, W, None, "bf16", tile, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ops.gemm_probe(A, W, None, "bf16", tile, n + 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    ops.gemm_probe(A, W, None, "bf16", tile, 1); torch.cuda.synchronize(); t2 = time.perf_counter()
    return ((t1 - t0) - (t2 - t1)) / n * 1e6
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
M = 778 * a.batch
for name, N, K in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    if a.baseline:
        os.environ["D2S_SPLITK_FORCE"] = "0"; ops.reload_env()
        print(f"{name:5s} M={M} N={N} K={K}: engine's choice (tile rule, lean ring) {t(A, W, 0, 300):6.1f} us", flush=True)
        continue
    for tile, (bm, bn) in ((3264, (32, 64)), (64648, (64, 64)), (641288, (64, 128)), (1281288, (128, 128))):
        row = []
        for ks in (1, 2, 4, 8):
            if (K // 64) % ks or K // 64 // ks < 2: row.append("   -  "); continue
            os.environ["D2S_SPLITK_FORCE"] = str(ks if ks > 1 else 0); ops.reload_env()
            us = t(A, W, tile, 300)
            units = -(-M // bm) * -(-N // bn) * ks
            row.append(f"{us:5.1f} ({units:4d} units{'' if ks == 1 else f', fused~{us - 6.1 + 2.5:5.1f}'})")
        print(f"{name:5s} tile {bm:3d}x{bn:3d}  K ranges 1 | 2 | 4 | 8:  " + "  |  ".join(row), flush=True)
