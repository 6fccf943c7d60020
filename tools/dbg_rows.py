import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from desktop2stereo_amd import ops, synth, _lib
lib = _lib.load(); dev = torch.device("cuda")
H, W = 1080, 1920
def run(B, mode, wpc=None):
    if wpc: os.environ["D2S_WARP_WPC"] = str(wpc)
    img = torch.from_numpy(np.stack([synth.noise_frame(H, W, i) for i in range(B)])).to(dev)
    dep = torch.from_numpy(np.stack([synth.smooth_depth(294, 518, i) for i in range(B)])).to(dev)
    sp = ops.sbs_params(0.064, 4.0, 0.0, mode, True)
    outs = {}
    for g in (0, 1):
        os.environ["D2S_WARP_GATHER"] = str(g); lib.d2s_debug_reload_env()
        outs[g] = ops.make_sbs(img, dep, sp).cpu().numpy().astype(np.int16)
    d = np.abs(outs[1] - outs[0]).max(axis=(-1))          # [B, oh, ow]
    bad_rows = np.where((d > 1).any(axis=-1))
    print(mode, "B", B, "wpc", wpc, "bad rows:", len(bad_rows[1]), "first:", bad_rows[1][:24], "frames", np.unique(bad_rows[0])[:8])
    if len(bad_rows[1]):
        r = bad_rows[1][0]; cols = np.where(d[bad_rows[0][0], r] > 1)[0]
        print("   row", r, "bad cols", len(cols), cols[:10], cols[-5:])
for wpc in (16, 4, 1):
    run(1, "Full-SBS", wpc)
run(2, "Half-TAB", 16)
