"""Correctness of the register-staged GEMM tiles against torch matmul (bf16 operands, fp32 accumulate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desktop2stereo_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
for (M, N, K) in [(778, 768, 768), (1000, 2304, 3072), (300, 64, 576), (5000, 32, 128)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    ref = A.bfloat16().float() @ W.bfloat16().float().T + b
    for tile in [256128, 128, 64, 3264, 1281288, 641288, 64648, 256648, 128324, 964, 91288, 912832, 9256648]:
        if tile in (925632,) and N > 32 and False: continue
        out = ops.gemm_probe(A, W, b, "bf16", tile, 1)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        print(M, N, K, tile, "rel err", f"{err:.2e}", "OK" if err < 2e-3 else "FAIL", flush=True)
    for tile in [91288, 964, 912832]:
        ref32 = A @ W.T + b
        out = ops.gemm_probe(A, W, b, "fp32", tile, 1)
        err = (out - ref32).abs().max().item() / ref32.abs().max().item()
        print(M, N, K, tile, "fp32 rel err", f"{err:.2e}", "OK" if err < 1e-5 else "FAIL", flush=True)
