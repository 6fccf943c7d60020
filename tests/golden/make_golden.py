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


def gen_model(model: str, res: int, frames, store_inputs: bool, full_taps: bool, out: str, fp32=True, metric="",
              cuda_branch=False, square=False, post_only=False):
    """predict_depth taps for one (model, depth_resolution).  cuda_branch: run _resize_patch_aligned_t's IS_CUDA branch
    (bicubic + antialias, reference depth.py:698-699 -- what the reference does on a CUDA *or ROCm* device; the flag is
    read at call time) instead of the CPU branch this container would take."""
    import torch
    from ref_harness import load_reference
    from desktop2stereo_amd import synth
    D = load_reference(model, res, seed=0, fp32=fp32, metric=metric)
    if cuda_branch:
        D.IS_CUDA = True
    captured = []
    if square:
        # get_patch_size() reads the module global at call time (depth.py:531-538); the model input predict_depth builds on this
        # branch is captured at the model call itself (a transparent proxy around the reference's wrapper object)
        D.CAPTURE_MODE = "Window"
        assert D.get_patch_size() is None
        inner = D.model_wraper

        class _Tap:
            def __call__(self, t):
                captured.append(t.detach().clone())
                return inner(t)

            def __getattr__(self, k):
                return getattr(inner, k)
        D.model_wraper = _Tap()
    data = {}
    meta = {"model": model, "depth_resolution": res, "weights_seed": 0, "fp32": fp32, "metric": metric, "cuda_branch": cuda_branch,
            "square": square,
            "max_depth": {"": 0.0, "Indoor": 20.0, "Outdoor": 80.0}[metric],
            "frames": [], "versions": _versions()}
    if metric:
        # normalize()'s metric branch on a map WITH invalid pixels (d <= 0): the order statistics run over the
        # compacted valid values (depth.py:844-847), which the model outputs (sigmoid > 0) never exercise
        assert D.is_metric()
        rng = np.random.default_rng(5)
        dm = (synth.smooth_depth(120, 200, 3) * 19.0 + 0.5).astype(np.float32)
        hole = rng.random(dm.shape) < 0.13
        dm[hole] = np.where(rng.random(int(hole.sum())) < 0.5, 0.0, -1.5).astype(np.float32)
        data["normcase_in"] = dm
        data["normcase_norm"] = D.normalize(torch.from_numpy(dm.copy())).numpy()
        data["normcase_post"] = D.post_process_depth(torch.from_numpy(dm.copy())).numpy()
        few = np.zeros((4, 8), np.float32)
        few.reshape(-1)[:7] = np.linspace(1, 7, 7)                      # 7 valid values (<= 10) -> bounds (0, 0)
        data["fewcase_in"] = few
        data["fewcase_norm"] = D.normalize(torch.from_numpy(few.copy())).numpy()
    D.depth_stabilizer.prev = None
    for fi, (kind, h, w, seed) in enumerate(frames):
        img = synth.structured_frame(h, w, seed) if kind == "S2" else synth.noise_frame(h, w, seed)
        meta["frames"].append({"kind": kind, "h": h, "w": w, "seed": seed})
        x = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
        if square:
            del captured[:]
            D.predict_depth(img, use_temporal_smooth=False)          # the reference builds the input itself (depth.py:1937-1950)
            xn = captured[-1].float()                                 # (earlier entries: the engine warm-up on zeros, depth.py:1861)
            assert tuple(xn.shape) == (1, 3, res, res), xn.shape
        else:
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
        if cuda_branch:
            data[pre + "resized_rows"] = xr[0, :, ::14].numpy()          # every 14th row of the resized (un-normalised) frame
        if square and not full_taps:
            data[pre + "model_input_rows"] = xn[0, :, ::37].numpy()
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
        if not (cuda_branch and fi > 0):                                  # (cuda_branch: depth for the first frame only)
            if not post_only:
                data[pre + "raw_depth"] = raw[0].float().numpy()
            data[pre + "post_depth"] = post.float().numpy()
        # end-to-end predict_depth with the EMA chain running across frames (depth.py:1983-1984)
        d_ema = D.predict_depth(img, use_temporal_smooth=True).float().numpy()
        if (full_taps or h * w <= 200 * 400) and not post_only:
            data[pre + "depth_ema_full"] = d_ema
            D2 = D.depth_stabilizer.prev
            data[pre + "ema_state"] = D2.float().numpy().copy()
        meta["frames"][-1]["raw_range"] = [float(raw.min()), float(raw.max())]
    np.savez_compressed(out + ".npz", **data)
    with open(out + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", out, {k: v.shape for k, v in data.items()})


def gen_warp(out: str, shapes=None, big=("hd",), s1_cases=(0, 1)):
    """make_sbs / make_sbs_core outputs with a GIVEN depth (isolates A14)."""
    import torch
    from ref_harness import load_reference
    from desktop2stereo_amd import synth
    D = load_reference("tiny", 84, seed=0, fp32=True)
    data, meta = {}, {"cases": [], "versions": _versions()}
    shapes = shapes or [("small169", 72, 128, 1), ("small43", 96, 128, 1), ("wide", 60, 160, 1),
                        ("odd", 75, 133, 1), ("hd", 1080, 1920, 135)]   # row stride of stored outputs
    for name, h, w, rs in shapes:
        for kind in ("S2", "S1"):
            if name not in big and kind == "S1":
                continue
            img = synth.structured_frame(h, w, 7) if kind == "S2" else synth.noise_frame(h, w, 7)
            dep = synth.smooth_depth(h, w, 7)
            if name not in big:
                data[f"{name}_{kind}_img"] = img
                data[f"{name}_{kind}_depth"] = dep
            for ci, (mode, fill, conv, ratio) in enumerate(WARP_CASES):
                if name in big and kind == "S1" and ci not in s1_cases:
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


def gen_warp_bf16(out: str):
    """make_sbs AS SHIPPED on the reference's CPU path: predict_depth returns a bf16 map there (CPU autocast, depth.py:661-664), and
    make_sbs casts rgb to the depth's dtype (depth.py:2209 / 2215) -- the whole warp (grid construction, grid_sample, area mean, clamp)
    then runs on bf16 tensors.  Same frames / smooth depth / cases as gen_warp's 1080p rows (the depth is rounded to bf16 first, as a
    bf16 predict_depth output would be; the fixture stores that rounded depth's seed, not the map).  Reported against, never gated:
    8 mantissa bits of colour cannot be '<= 1 LSB' of anything -- the point is to state how far the HIP fp32 warp is from it."""
    import torch
    from ref_harness import load_reference
    from desktop2stereo_amd import synth
    D = load_reference("tiny", 84, seed=0, fp32=False)
    data, meta = {}, {"cases": [], "versions": _versions(), "note": "depth = bf16(synth.smooth_depth(h, w, 7)); rgb cast to bf16 by make_sbs"}
    h, w, rs = 1080, 1920, 135
    for kind in ("S2", "S1"):
        img = synth.structured_frame(h, w, 7) if kind == "S2" else synth.noise_frame(h, w, 7)
        dep = torch.from_numpy(synth.smooth_depth(h, w, 7)).to(torch.bfloat16)
        for ci, (mode, fill, conv, ratio) in enumerate(WARP_CASES):
            if kind == "S1" and ci not in (1, 5):
                continue
            sbs = D.make_sbs(img, dep, ipd_uv=0.064, depth_ratio=ratio, convergence=conv, fill_16_9=fill, display_mode=mode)
            assert sbs.dtype == np.float32
            key = f"hd_{kind}_c{ci}"
            data[key] = np.rint(sbs[::rs] * 256.0).astype(np.uint16)
            meta["cases"].append({"key": key, "h": h, "w": w, "kind": kind, "seed": 7, "mode": mode, "fill_16_9": fill, "convergence": conv,
                                  "depth_ratio": ratio, "ipd_uv": 0.064, "row_stride": rs, "out_shape": list(sbs.shape)})
    np.savez_compressed(out + ".npz", **data)
    with open(out + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", out, len(data), "arrays")


INGEST_CASES = [  # (name, H0, W0, channels, target_height, stored row stride)
    ("bgra_1080_to_720", 1080, 1920, 4, 720, 45), ("bgr_270_to_100", 270, 480, 3, 100, 1),
    ("bgr_90_keep", 90, 160, 3, 90, 1), ("bgra_101_to_33", 101, 75, 4, 33, 1), ("bgr_2160_to_1080", 2160, 3840, 3, 1080, 120),
]
OVERLAY_CASES = [("h90", 90, 160, 59.9), ("h270", 270, 480, 123.4), ("h1080", 1080, 1920, 7.0), ("narrow", 120, 30, 60.0)]


def _reference_process_defs(D):
    """Both definitions of process() in the reference (`if IS_CUDA: def process ... else: def process ...`, depth.py:540-629):
    only one exists after import, so the two FunctionDef nodes are taken from the reference's source with `ast` at generation time
    and executed in the imported module's own namespace (DEVICE = cpu, DTYPE = float32).  Nothing of the source is stored."""
    import ast
    import torch
    with open(os.path.join("/root/reference", "depth.py")) as f:
        tree = ast.parse(f.read())
    node = next(n for n in tree.body if isinstance(n, ast.If) and isinstance(n.test, ast.Name) and n.test.id == "IS_CUDA"
                and any(isinstance(c, ast.FunctionDef) and c.name == "process" for c in n.body))
    fns = {}
    for key, body in (("cuda", node.body), ("cpu", node.orelse)):
        fd = next(c for c in body if isinstance(c, ast.FunctionDef) and c.name == "process")
        ns = dict(D.__dict__)
        ns.update(DEVICE=torch.device("cpu"), DTYPE=torch.float32)
        exec(compile(ast.Module(body=[fd], type_ignores=[]), f"<reference depth.py:{fd.lineno}-{fd.end_lineno}>", "exec"), ns)
        fns[key] = (ns["process"], [fd.lineno, fd.end_lineno])
    return fns


TENSOR_CASES = [  # (name, layout, planes/channels, dtype, H0, W0, target_height, stored row stride): process()'s tensor branch
    ("chw3_u8_270_to_100", "chw", 3, "u8", 270, 480, 100, 1), ("chw4_u8_101_to_33", "chw", 4, "u8", 101, 75, 33, 1),
    ("hwc4_u8_1080_to_720", "hwc", 4, "u8", 1080, 1920, 720, 45), ("chw3_f32_90_to_41", "chw", 3, "f32", 90, 160, 41, 1),
    ("chw3_u8_90_keep", "chw", 3, "u8", 90, 160, 90, 1),
]


def gen_ingest(out: str):
    """A1 process() and A15 overlay_fps().  process(): BOTH of the reference's definitions are executed (see
    _reference_process_defs): the IS_CUDA one (depth.py:540-566, what a ROCm device runs) on BGR(A) uint8 frames, and the tensor
    branch of the other (depth.py:576-601).  Its cv2 branch (INTER_AREA) cannot run here: cv2 is not installed.
    overlay_fps() is the reference's own function."""
    import torch
    from ref_harness import load_reference
    from desktop2stereo_amd import synth
    D = load_reference("tiny", 84, seed=0, fp32=True)
    fns = _reference_process_defs(D)
    data, meta = {}, {"process": [], "process_tensor": [], "overlay": [], "versions": _versions(),
                      "process_source": {k: f"reference depth.py:{v[1][0]}-{v[1][1]}, executed via ast extraction" for k, v in fns.items()}}
    for name, H0, W0, ch, target, rs in INGEST_CASES:
        img = np.random.default_rng([11, H0, W0]).integers(0, 256, (H0, W0, ch), dtype=np.uint8)
        res = fns["cuda"][0](img.copy(), target).float().numpy()
        data["process_" + name] = res[:, ::rs]
        meta["process"].append({"name": name, "H0": H0, "W0": W0, "channels": ch, "target": target, "row_stride": rs,
                                "out_shape": list(res.shape), "seed": [11, H0, W0]})
    for name, layout, ch, dt, H0, W0, target, rs in TENSOR_CASES:
        rng = np.random.default_rng([12, H0, W0])
        shape = (ch, H0, W0) if layout == "chw" else (H0, W0, ch)
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        t = torch.from_numpy(img) if dt == "u8" else torch.from_numpy(img.astype(np.float32) + np.float32(0.25))
        r = fns["cpu"][0](t.clone(), target)
        res = r.numpy()
        data["ptensor_" + name] = res[:, ::rs]
        meta["process_tensor"].append({"name": name, "layout": layout, "channels": ch, "dtype": dt, "H0": H0, "W0": W0, "target": target,
                                       "row_stride": rs, "out_shape": list(res.shape), "out_dtype": str(res.dtype), "seed": [12, H0, W0]})
    for name, H, W, fps in OVERLAY_CASES:
        rgb = torch.from_numpy(synth.structured_frame(H, W, 4)).permute(2, 0, 1).float()
        D._FPS_MASK_CACHE.update(mask=None, frame=0)
        res = D.overlay_fps(rgb, fps).numpy()
        box_h, box_w = min(H, 60), min(W, 420)
        data["overlay_" + name] = res[:, :box_h, :box_w]
        meta["overlay"].append({"name": name, "H": H, "W": W, "fps": fps, "seed": 4, "box": [box_h, box_w],
                                "changed_outside_box": bool((res[:, box_h:] != rgb.numpy()[:, box_h:]).any()
                                                            or (res[:, :, box_w:] != rgb.numpy()[:, :, box_w:]).any())})
    np.savez_compressed(out + ".npz", **data)
    with open(out + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", out, {k: v.shape for k, v in data.items()})


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
    # the reference AS SHIPPED (bf16 CPU autocast) on the other full-size fixtures: post-processed depth only -- the bound the HIP bf16
    # engine is graded against is the reference's own bf16-vs-fp32 distance on the same frame
    "tiny_r518_bf16": lambda o: gen_model("tiny", 518, [("S2", 1080, 1920, 0)], False, False, o, fp32=False, post_only=True),
    "vits_r336_bf16": lambda o: gen_model("vits", 336, [("S2", 1080, 1920, 0)], False, False, o, fp32=False, post_only=True),
    "vitl_r518_4k_bf16": lambda o: gen_model("vitl", 518, [("S2", 2160, 3840, 0)], False, False, o, fp32=False, post_only=True),
    "vitb_r518_bf16": lambda o: gen_model("vitb", 518, [("S2", 1080, 1920, 0)], False, False, o, fp32=False),
    # config 3 shapes: ViT-L, 3840x2160 frame (CPU branch decimates ::3 before the bilinear resize)
    "vitl_r518_4k": lambda o: gen_model("vitl", 518, [("S2", 2160, 3840, 0)], False, False, o),
    # metric head + metric normalize (reference ids Depth-Anything-V2-Metric-{Indoor,Outdoor}-*)
    "tiny_r84_metric": lambda o: gen_model("tiny", 84, [("S2", 90, 160, 0), ("S2", 90, 160, 1), ("S1", 90, 160, 2)],
                                           True, True, o, metric="Indoor"),
    "vits_r518_metric": lambda o: gen_model("vits", 518, [("S2", 1080, 1920, 0)], False, False, o, metric="Outdoor"),
    # the IS_CUDA pre-processing branch (bicubic + antialias from the full frame): 1080p, 4K, 1440p, 720p and an odd size
    "vits_r518_cuda": lambda o: gen_model("vits", 518, [("S2", 1080, 1920, 0), ("S1", 2160, 3840, 1), ("S2", 1440, 2560, 2),
                                                        ("S1", 720, 1280, 3), ("S2", 611, 1003, 4)], False, False, o, cuda_branch=True),
    "warp": gen_warp,
    "warp_bf16": gen_warp_bf16,
    # __graft_entry__.smoke()'s frame (KAT model, 270 x 480, Depth Resolution 140): the reference's fp32 result and its as-shipped bf16
    # result -- smoke's bf16 tolerance is their distance, not a guess
    "smoke_tiny_r140": lambda o: gen_model("tiny", 140, [("S2", 270, 480, 0)], False, False, o, post_only=True),
    "smoke_tiny_r140_bf16": lambda o: gen_model("tiny", 140, [("S2", 270, 480, 0)], False, False, o, fp32=False, post_only=True),
    # BASELINE config 3's frame: 3840x2160, all four packings incl. Full-TAB 4320x3840 / Half-TAB (every 270th row stored;
    # noise frame: Full-TAB and Half-TAB only)
    "warp_uhd": lambda o: gen_warp(o, shapes=[("uhd", 2160, 3840, 270)], big=("uhd",), s1_cases=(2, 3)),
    "ingest": gen_ingest,
    # the fixed-square input branch of predict_depth (CAPTURE_MODE == "Window" -> get_patch_size() is None, depth.py:531-538,
    # 1937-1946): every tap of the KAT model at 84 x 84, and ViT-S at 518 x 518 (1 370 tokens) from a 1080p frame
    "tiny_r84_square": lambda o: gen_model("tiny", 84, [("S2", 90, 160, 0), ("S1", 84, 84, 1)], True, True, o, square=True),
    "vits_r518_square": lambda o: gen_model("vits", 518, [("S2", 1080, 1920, 0)], False, False, o, square=True),
}

if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--job":
        JOBS[sys.argv[2]](os.path.join(HERE, sys.argv[2]))
    else:
        names = sys.argv[1:] or list(JOBS)
        for n in names:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--job", n])
