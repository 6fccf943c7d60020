"""Phase times of conv3_head_ups_kernel (the head's conv2 with the up-sample folded in, producer / consumer waves).
Build the instrumented variant first:   tools/build_variant.sh c3u conv3.hip -DD2S_C3U_TIMING
    D2S_LIB=desktop2stereo_amd/libd2s_hip_c3u.so python tools/c3u_timeline.py [B]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from desktop2stereo_amd import _lib, ops
from desktop2stereo_amd.config import MODELS, engine_shape
from desktop2stereo_amd.weights import make_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = C.CDLL(_lib.LIB_PATH)
lib.d2s_c3u_timing.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda", 0)
cfg = MODELS["vitb"]
h, w, _ = engine_shape(1080, 1920, 518)
eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, B, "bf16")
x = torch.randn(B, 3, h, w, device=dev)
for _ in range(3):
    eng(x)
torch.cuda.synchronize()
lib.d2s_c3u_timing(None, 1)
eng(x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 9))()
lib.d2s_c3u_timing(buf, 0)
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 9).astype(np.float64)
t = t[t[:, 8] > 0]
tiles = t[:, 8]
print(f"{len(t)} blocks, {tiles.mean():.1f} tiles per block")
for name, k in (("consumer: input rows 0-2 (72 MFMAs)", 0), ("consumer: wait at barrier 1", 1), ("consumer: rows 3-5 + epilogue", 2), ("consumer: wait at barrier 2", 3),
                ("producer: H pass (compute)", 4), ("producer: wait at barrier 1", 5), ("producer: H requests + V pass", 6), ("producer: wait at barrier 2", 7)):
    per = t[:, k] / tiles / 100.0
    print(f"   {name:40s} {per.mean():6.2f} us per tile   (min {per.min():.2f}, max {per.max():.2f})")
