#!/usr/bin/env python
"""Micro-benchmark of the MFMA GEMM kernel on the engine's shapes (d2s_gemm_probe).
    python tools/gemm_bench.py [--batch 1 16] [--prec bf16]"""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desktop2stereo_amd import ops

def t_probe(A, W, prec, tile, iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.gemm_probe(A, W, None, prec, tile, iters)
    torch.cuda.synchronize(); return time.perf_counter() - t0

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, nargs="+", default=[1, 16]); ap.add_argument("--prec", default="bf16")
ap.add_argument("--tiles", type=int, nargs="+", default=[0, 3264, 64, 64648, 641288, 1281288, 91288, 256128])
ap.add_argument("--check", action="store_true", help="also compare every tile code with a float64 product of the bf16-rounded operands")
a = ap.parse_args()
dev = torch.device("cuda")
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
for B in a.batch:
    for name, N, K in shapes:
        M = 778 * B
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
        want = (A.bfloat16().double() @ W.bfloat16().double().T).float() if a.check and a.prec == "bf16" else None
        for tile in a.tiles:
            if want is not None:
                err = ((ops.gemm_probe(A, W, None, a.prec, tile, 1) - want).abs().max() / want.abs().max()).item()
                if err > 2e-5: print(f"   tile {tile}: WRONG (max rel err {err:.2e})", flush=True)
            t_probe(A, W, a.prec, tile, 3)
            n = 200 if B == 1 else 50
            dt = (t_probe(A, W, a.prec, tile, n + 1) - t_probe(A, W, a.prec, tile, 1)) / n
            print(f"B={B:3d} {name:5s} M={M:6d} N={N:5d} K={K:5d} tile={tile:8d}: {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:7.1f} TF/s", flush=True)
