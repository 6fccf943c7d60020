"""Phase times of conv3_wide_kernel (build with D2S_HIPCC_DEFS=-DD2S_C3_TIMING, `rm desktop2stereo_amd/csrc/_build/conv3.o` first).
Runs one batch-B ViT-B forward and prints, summed over the wide-kernel launches of the step and averaged over blocks, the time per
tile spent in the K loop / epilogue / barrier + halo store.      python tools/c3_timeline.py [B]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from desktop2stereo_amd import _lib, ops, synth
from desktop2stereo_amd.config import MODELS, engine_shape
from desktop2stereo_amd.weights import make_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = C.CDLL(_lib.LIB_PATH)
lib.d2s_c3_timing.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda", 0)
cfg = MODELS["vitb"]
h, w, _ = engine_shape(1080, 1920, 518)
eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, B, "bf16")
x = torch.randn(B, 3, h, w, device=dev)
for _ in range(3):
    eng(x)
torch.cuda.synchronize()
lib.d2s_c3_timing(None, 1)
eng(x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 8))()
lib.d2s_c3_timing(buf, 0)
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
t = t[t[:, 3] > 0]
tiles = t[:, 3]
print(f"{len(t)} blocks, {int(tiles.sum())} tiles in the step's wide launches ({tiles.mean():.1f} per block)")
for name, k in (("K loop", 0), ("  of it K tile 0 (drain + first wait)", 4), ("epilogue: residual requests", 1), ("barrier + halo store", 2), ("epilogue: arithmetic + stores", 6)):
    per = t[:, k] / tiles / 100.0
    print(f"   {name:40s} {per.mean():6.2f} us per tile   (min {per.min():.2f}, max {per.max():.2f})")
print(f"   block lifetime, all launches               {t[:, 5].mean() / 100.0:8.1f} us;  accounted {(t[:, 0] + t[:, 1] + t[:, 2] + t[:, 6]).mean() / 100.0:8.1f} us")
print("   ideal K loop: 18 K tiles x 2 waves x 32 MFMA x 16 cycles = 18 432 cycles = 7.7 us at 2.4 GHz")
