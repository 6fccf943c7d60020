"""Attention kernel alone (d2s_attention_probe): TFLOP/s at the engine's shapes, both kernels.
usage (GPU box): python tools/attn_bench.py [B ...]      D2S_ATTN32=0 selects the 16-row kernel"""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from desktop2stereo_amd import ops

def main():
    batches = [int(a) for a in sys.argv[1:]] or [1, 8, 16, 27, 32]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    for heads, N in ((12, 778), (16, 1370)):
        for B in batches:
            q, k, v = (torch.randn((B, heads, N, 64), generator=g).to(dev) for _ in range(3))
            best = 1e9
            for _ in range(3):
                _, ms = ops.attention_probe(q, k, v, "bf16", iters=21)
                best = min(best, ms)
            fl = 4.0 * B * heads * N * N * 64
            print(f"heads {heads} N {N} B {B}: {best * 1e3:8.1f} us  {fl / (best * 1e-3) * 1e-12:7.1f} TFLOP/s  ({fl / (best * 1e-3) / 2.5e15:.3f} of peak)", flush=True)

if __name__ == "__main__":
    main()
