#!/usr/bin/env python
"""Correctness + race screen + timing of the 256x256 ping-pong GEMM (tile code 256256 / 256256<k> = K split k) through
d2s_gemm_probe, against a float64 reference of the bf16-rounded operands.
    python tools/pp_check.py [--bench] [--batch 16 32]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desktop2stereo_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--bench", action="store_true")
ap.add_argument("--batch", type=int, nargs="+", default=[16, 32])
ap.add_argument("--prec", default="bf16")
ap.add_argument("--tiles", type=int, nargs="+", default=[0, 256256])
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)


def ref(A, W, bias, prec):
    if prec == "bf16":
        A, W = A.bfloat16().double(), W.bfloat16().double()
    else:
        A, W = A.to(torch.float8_e4m3fn).double(), W.to(torch.float8_e4m3fn).double()
    return (A @ W.T + bias.double()).float()


bad = 0
for (M, N, K) in [(700, 512, 256), (1000, 768, 768), (3112, 2304, 768), (3112, 768, 3072), (12448, 3072, 768), (24896, 768, 768), (513, 1024, 512), (2000, 256, 1024)]:
    A = torch.randn(M, K, device=dev) * 0.5
    W = torch.randn(N, K, device=dev) * 0.5
    A[:, 0] += torch.arange(M, device=dev) % 7 * 0.25            # asymmetric: catches transposes / row permutations
    W[:, 1] += torch.arange(N, device=dev) % 5 * 0.25
    bias = torch.randn(N, device=dev)
    want = ref(A, W, bias, a.prec)
    scale = want.abs().max().item()
    first = None
    for rep in range(8):                                          # race screen: every run must be bit-identical
        got = ops.gemm_probe(A, W, bias, a.prec, 256256, 1)
        if first is None:
            first = got.clone()
            err = (got - want).abs().max().item() / scale
            ok = err <= (2e-5 if a.prec == "bf16" else 5e-5)        # (fp32 accumulation order; e4m3 operands of this spread leave 2-3e-5)
            print(f"M={M:6d} N={N:5d} K={K:5d}  max rel err {err:.2e}  {'ok' if ok else 'WRONG'}", flush=True)
            bad += not ok
        elif not torch.equal(got, first):
            print(f"   run {rep}: differs from run 0 by {(got - first).abs().max().item():.3e}  RACE", flush=True)
            bad += 1
print("FAILED" if bad else "all correct, runs bit-identical")

if a.bench:
    def t_probe(A, W, prec, tile, iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.gemm_probe(A, W, None, prec, tile, iters)
        torch.cuda.synchronize(); return time.perf_counter() - t0
    shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
    for B in a.batch:
        for name, N, K in shapes:
            M = 778 * B
            A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
            for tile in a.tiles:
                t_probe(A, W, a.prec, tile, 3)
                n = 50
                best = min((t_probe(A, W, a.prec, tile, n + 1) - t_probe(A, W, a.prec, tile, 1)) / n for _ in range(3))
                print(f"B={B:3d} {name:5s} M={M:6d} N={N:5d} K={K:5d} tile={tile:8d}: {best*1e6:8.1f} us  {2*M*N*K/best/1e12:7.1f} TF/s", flush=True)
