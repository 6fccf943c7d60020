import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from desktop2stereo_amd import ops
dev = torch.device("cuda")
def t_probe(A, W, tile, iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.gemm_probe(A, W, None, "bf16", tile, iters)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for (M, N, K) in [(4096, 4096, 768), (4096, 4096, 3072), (4096, 4096, 12288), (8192, 8192, 768), (8192, 8192, 8192), (12448, 2304, 768), (12448, 768, 3072)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    for tile in (0, 256256):
        t_probe(A, W, tile, 3)
        n = 30
        best = min((t_probe(A, W, tile, n + 1) - t_probe(A, W, tile, 1)) / n for _ in range(3))
        print(f"M={M} N={N} K={K} tile={tile}: {best*1e6:8.1f} us {2*M*N*K/best/1e12:7.1f} TF/s", flush=True)
