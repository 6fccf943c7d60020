"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (it imports /root/reference through ref_harness.py).  Each
configuration needs its own process because the reference fixes MODEL / DEPTH_RESOLUTION at
import time (depth.py:1784, utils.py:834-837):

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py tiny_r84   # one

Inputs are regenerated from seeds (desktop2stereo_amd.synth) and are stored only when tiny.
Fixtures are data (inputs + the reference's outputs); no reference source is stored.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

WARP_CASES = [  # (mode, fill_16_9, convergence, depth_ratio)
    ("Half-SBS", True, 0.0, 4.0), ("Full-SBS", True, 0.0, 4.0), ("Half-TAB", True, 0.0, 4.0),
    ("Full-TAB", True, 0.0, 4.0), ("Half-SBS", False, 0.1, 2.0), ("Full-SBS", False, 0.1, 2.0),
    ("Half-TAB", False, 0.1, 2.0), ("Full-TAB", False, 0.1, 2.0),
]


def _versions():
    import torch
    import transformers
    return {"torch": torch.__version__, "transformers": transformers.__version__,
            "numpy": np.__version__, "reference_pins": "transformers==4.56.2, torch 2.7.1"}


def gen_model(model: str, res: int, frames, store_inputs: bool, full_taps: bool, out: str, fp32=True):
    """predict_depth taps for one (model, depth_resolution)."""
    import torch
    from ref_harness import load_reference
    from desktop2stereo_amd import synth
    D = load_reference(model, res, seed=0, fp32=fp32)
    data = {}
    meta = {"model": model, "depth_resolution": res, "weights_seed": 0, "fp32": fp32,
            "frames": [], "versions": _versions()}
    D.depth_stabilizer.prev = None
    for fi, (kind, h, w, seed) in enumerate(frames):
        img = synth.structured_frame(h, w, seed) if kind == "S2" else synth.noise_frame(h, w, seed)
        meta["frames"].append({"kind": kind, "h": h, "w": w, "seed": seed})
        x = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
        xr = D._resize_patch_aligned_t(x, res, 14)
        xn = xr / 255.0
        m, s = D._normalization_tensors_for(xn)
        xn = (xn - m) / s
        with torch.no_grad():
            raw = D.model_wraper(xn)
            post = D.post_process_depth(raw.float())
        pre = f"f{fi}_"
        if store_inputs:
            data[pre + "img"] = img
        if full_taps:
            data[pre + "model_input"] = xn[0].numpy()
            with torch.no_grad():
                bo = D.model_wraper.model.backbone(xn, output_hidden_states=True)
            for li, hs in enumerate(bo.hidden_states):
                data[pre + ("embeddings" if li == 0 else f"layer{li}")] = hs[0].float().numpy()
            nrm = D.normalize(raw.float())
            data[pre + "norm"] = nrm.numpy()
            data[pre + "gamma"] = D.apply_gamma(nrm).numpy()
            data[pre + "fg"] = D.apply_foreground_scale(D.apply_gamma(nrm), D.FOREGROUND_SCALE).numpy()
        data[pre + "raw_depth"] = raw[0].float().numpy()
        data[pre + "post_depth"] = post.float().numpy()
        # end-to-end predict_depth with the EMA chain running across frames (depth.py:1983-1984)
        d_ema = D.predict_depth(img, use_temporal_smooth=True).float().numpy()
        if full_taps or h * w <= 200 * 400:
            data[pre + "depth_ema_full"] = d_ema
            D2 = D.depth_stabilizer.prev
            data[pre + "ema_state"] = D2.float().numpy().copy()
        meta["frames"][-1]["raw_range"] = [float(raw.min()), float(raw.max())]
    np.savez_compressed(out + ".npz", **data)
    with open(out + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", out, {k: v.shape for k, v in data.items()})


def gen_warp(out: str):
    """make_sbs / make_sbs_core outputs with a GIVEN depth (isolates A14)."""
    import torch
    from ref_harness import load_reference
    from desktop2stereo_amd import synth
    D = load_reference("tiny", 84, seed=0, fp32=True)
    data, meta = {}, {"cases": [], "versions": _versions()}
    shapes = [("small169", 72, 128, 1), ("small43", 96, 128, 1), ("wide", 60, 160, 1),
              ("odd", 75, 133, 1), ("hd", 1080, 1920, 135)]             # row stride of stored outputs
    for name, h, w, rs in shapes:
        for kind in ("S2", "S1"):
            if name != "hd" and kind == "S1":
                continue
            img = synth.structured_frame(h, w, 7) if kind == "S2" else synth.noise_frame(h, w, 7)
            dep = synth.smooth_depth(h, w, 7)
            if name != "hd":
                data[f"{name}_{kind}_img"] = img
                data[f"{name}_{kind}_depth"] = dep
            for ci, (mode, fill, conv, ratio) in enumerate(WARP_CASES):
                if name == "hd" and kind == "S1" and ci > 1:
                    continue
                sbs = D.make_sbs(img, torch.from_numpy(dep), ipd_uv=0.064, depth_ratio=ratio,
                                 convergence=conv, fill_16_9=fill, display_mode=mode)
                key = f"{name}_{kind}_c{ci}"
                # every rs-th row, as uint16 fixed point (value*256: 1/256 resolution, well below
                # the reference's own float32 coordinate noise of ~0.04)
                data[key] = np.rint(sbs[::rs] * 256.0).astype(np.uint16)
                meta["cases"].append({"key": key, "shape": name, "h": h, "w": w, "kind": kind, "seed": 7,
                                      "mode": mode, "fill_16_9": fill, "convergence": conv,
                                      "depth_ratio": ratio, "ipd_uv": 0.064, "row_stride": rs,
                                      "out_shape": list(sbs.shape)})
    np.savez_compressed(out + ".npz", **data)
    with open(out + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", out, len(data), "arrays")


JOBS = {
    # KAT-tiny: every tap, 3 frames (EMA chain), inputs stored
    "tiny_r84": lambda o: gen_model("tiny", 84, [("S2", 90, 160, 0), ("S2", 90, 160, 1), ("S1", 90, 160, 2)],
                                    True, True, o),
    "tiny_r518": lambda o: gen_model("tiny", 518, [("S2", 1080, 1920, 0)], False, False, o),
    "vits_r518": lambda o: gen_model("vits", 518, [("S2", 1080, 1920, 0)], False, False, o),
    "vits_r336": lambda o: gen_model("vits", 336, [("S2", 1080, 1920, 0)], False, False, o),
    "vitb_r518": lambda o: gen_model("vitb", 518, [("S2", 1080, 1920, 0)], False, False, o),
    # the as-shipped CPU autocast (bf16) result, to report distance to it
    "vits_r518_bf16": lambda o: gen_model("vits", 518, [("S2", 1080, 1920, 0)], False, False, o, fp32=False),
    # config 3 shapes: ViT-L, 3840x2160 frame (CPU branch decimates ::3 before the bilinear resize)
    "vitl_r518_4k": lambda o: gen_model("vitl", 518, [("S2", 2160, 3840, 0)], False, False, o),
    "warp": gen_warp,
}

if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--job":
        JOBS[sys.argv[2]](os.path.join(HERE, sys.argv[2]))
    else:
        names = sys.argv[1:] or list(JOBS)
        for n in names:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--job", n])
