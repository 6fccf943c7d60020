"""e4m3 GEMM (d2s_gemm_probe, precision fp8, unit scales) against an emulation: operands rounded to OCP e4m3fn by
torch's float8_e4m3fn cast on the host, float32 matmul.  Also times the tiles against bf16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desktop2stereo_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
for (M, N, K) in [(778, 768, 768), (1000, 2304, 3072), (333, 128, 256)]:
    A = torch.randn(M, K) * 3; W = torch.randn(N, K); b = torch.randn(N)
    A[0, :8] = torch.tensor([500., -500., 448., 1e-3, 2e-3, 0.0175, -0.0009, 464.])       # saturation / subnormals
    Aq = A.to(torch.float8_e4m3fn).float(); Wq = W.to(torch.float8_e4m3fn).float()
    Aq = torch.where(A.abs() >= 448, torch.sign(A) * 448, Aq)                                # saturating cast (torch gives NaN)
    ref = Aq.double() @ Wq.double().T + b.double()
    for tile in [64, 3264, 964, 91288, 964128]:
        out = ops.gemm_probe(A.to(dev), W.to(dev), b.to(dev), "fp8", tile, 1).cpu().double()
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        print(M, N, K, tile, "rel err", f"{err:.2e}", "OK" if err < 1e-5 else "FAIL", flush=True)

def t_probe(A, W, prec, tile, iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.gemm_probe(A, W, None, prec, tile, iters)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for B in (1, 16):
    for name, N, K in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
        M = 778 * B
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
        for prec in ("bf16", "fp8"):
            t_probe(A, W, prec, 0, 3)
            n = 200 if B == 1 else 50
            dt = (t_probe(A, W, prec, 0, n + 1) - t_probe(A, W, prec, 0, 1)) / n
            print(f"B={B:3d} {name:5s} {prec}: {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:7.1f} TF/s", flush=True)
