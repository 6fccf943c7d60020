#!/usr/bin/env python
"""Calibration only (never on the product path): what the vendor library (hipBLASLt through torch.matmul / F.linear) reaches on the
batched encoder-linear shapes, next to the engine's own kernels (d2s_gemm_probe).  Tells how far the shapes themselves
(K = 768: 12 K tiles per output tile) are from the dense peak for a mature kernel.
    python tools/blaslt_calib.py [--batch 32]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[32])
ap.add_argument("--ours", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda")
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e-3


for B in a.batch:
    M = 778 * B
    for name, N, K in shapes:
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        W = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev, dtype=torch.bfloat16)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t1 = timeit(lambda: torch.matmul(A, W.T, out=C))
        t2 = timeit(lambda: F.linear(A, W, bias))
        print(f"B={B:3d} {name:5s} M={M:6d} N={N:5d} K={K:5d} hipBLASLt matmul {t1*1e6:7.1f} us {2*M*N*K/t1/1e12:7.1f} TF/s | linear+bias {t2*1e6:7.1f} us {2*M*N*K/t2/1e12:7.1f} TF/s", flush=True)
        if a.ours:
            from desktop2stereo_amd import ops
            Af, Wf = A.float(), W.float()
            def t_probe(iters):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ops.gemm_probe(Af, Wf, None, "bf16", 256256, iters)
                torch.cuda.synchronize(); return time.perf_counter() - t0
            t_probe(3)
            t3 = min((t_probe(51) - t_probe(1)) / 50 for _ in range(3))
            print(f"                                              gemm_pp (bf16 out, no epilogue work) {t3*1e6:7.1f} us {2*M*N*K/t3/1e12:7.1f} TF/s", flush=True)
    # big square for reference
for n in (4096, 8192):
    A = torch.randn(n, n, device=dev, dtype=torch.bfloat16); W = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    C = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: torch.matmul(A, W.T, out=C), 20)
    print(f"square {n}: hipBLASLt {t*1e6:8.1f} us {2*n**3/t/1e12:7.1f} TF/s", flush=True)
