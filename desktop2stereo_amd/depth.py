"""Drop-in call surface of the reference's ``depth.py`` for the hot path, backed by libd2s_hip.so.

Same names, argument meaning and return conventions as the reference (callers: reference
main.py:44, 244, 249, 1321, 1340):

    process(img, height)                                              reference depth.py:540-629
    predict_depth(image_rgb, return_tuple=False, use_temporal_smooth=True, dtype=None)   :1897-2025
    make_sbs(rgb_c, depth, ipd_uv=0.064, depth_ratio=2.0, convergence=0.0,
             fill_16_9=False, display_mode="Half-SBS", fps=None)      :2186-2231
    make_sbs_core(rgb, depth, ipd_uv, depth_ratio, display_mode, fill_16_9, convergence, device)  :2122-2184
    post_process_depth / DepthStabilizer / depth_stabilizer           :806-814, 1865-1889
plus ``predict`` / ``to_stereo`` (the names BASELINE.json's north_star uses) and the batched
``pipeline``.  Differences from the reference, all deliberate:

  * the module is configured explicitly (``configure``) instead of reading settings.yaml and
    building the model at import time (reference depth.py:1784; utils.py:635);
  * every computation is a HIP kernel behind the C-ABI; if the library or a GPU is missing the
    call raises -- there is no eager-PyTorch fallback (the reference degrades silently,
    depth.py:1597-1631);
  * ``process`` has both definitions of the reference behind ``branch=``: the torch one a ROCm device takes
    (depth.py:540-566, default) and the non-CUDA one (depth.py:570-629: tensor branch + cv2 INTER_AREA branch);
    only the cv2.UMat container of the latter is not built.
"""
from __future__ import annotations

import dataclasses
from threading import Lock
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops
from .config import METRIC_MODEL_IDS, MODELS, MODEL_IDS, ModelConfig, PipelineParams, engine_shape, is_metric_id

_state = {"cfg": None, "weights": None, "params": PipelineParams(), "precision": "bf16", "device": 0,
          "engine": None, "engine_key": None, "max_batch": 1}
lock = Lock()                         # one-time engine build (reference depth.py:1838-1862)


def configure(model="vitb", weights: Optional[Dict[str, np.ndarray]] = None, params: Optional[PipelineParams] = None,
              precision: str = "bf16", device: int = 0, max_batch: int = 1, seed: int = 0):
    """Select model / weights / constants (what settings.yaml + utils.py do in the reference).
    `model`: "vits" | "vitb" | "vitl" | "tiny" | a reference MODEL_ID | a ModelConfig.
    `weights`: HF-keyed float arrays, a path to model.safetensors, or None (seeded synthetic)."""
    from .weights import load_safetensors, make_weights
    # Video-Depth-Anything ids (reference utils.py model map; depth.py:875-893): streaming temporal head, one stream
    vda = {"depth-anything/Video-Depth-Anything-Small": "vits", "depth-anything/Video-Depth-Anything-Base": "vitb",
           "depth-anything/Video-Depth-Anything-Large": "vitl", "vda_tiny": "tiny", "vda_vits": "vits", "vda_vitb": "vitb",
           "vda_vitl": "vitl"}
    vda.update({k.replace("/Video-", "/Metric-Video-"): v for k, v in list(vda.items()) if "/Video-" in k})
    # metric ids (reference depth.py:666: is_metric() is keyed off the id): normalize() inverts 1/d; the HF
    # Metric-Indoor/Outdoor checkpoints additionally end in sigmoid * max_depth
    metric = isinstance(model, str) and is_metric_id(model)
    max_depth = 0.0
    if isinstance(model, str) and model in METRIC_MODEL_IDS:
        model, max_depth = METRIC_MODEL_IDS[model]
    temporal = isinstance(model, str) and model in vda
    if temporal:
        model = vda[model]
    cfg = model if isinstance(model, ModelConfig) else MODELS[MODEL_IDS.get(model, model)]
    if metric:
        params = dataclasses.replace(params or PipelineParams(), metric=True)
    if weights is None:
        if temporal:
            from .vda_weights import make_vda_weights
            weights = make_vda_weights(cfg, seed)
        else:
            weights = make_weights(cfg, seed)
    elif isinstance(weights, str):
        if temporal:                                    # reference .pth layout (pretrained.* / head.*) -> engine names
            from .vda_weights import vda_to_hf
            sd = torch.load(weights, map_location="cpu", weights_only=True)
            weights = vda_to_hf({k: v.float().numpy() for k, v in sd.items()}, cfg)
        else:
            weights = load_safetensors(weights, cfg)
    if temporal and max_batch != 1:
        raise _lib.D2SError("a Video-Depth-Anything engine is one stream: max_batch must be 1")
    with lock:
        if _state["engine"] is not None:
            _state["engine"].close()
        _state.update(cfg=cfg, weights=weights, params=params or PipelineParams(), precision=precision, device=device,
                      engine=None, engine_key=None, max_batch=max_batch, temporal=temporal, max_depth=max_depth)
    depth_stabilizer.prev = None


def _device() -> torch.device:
    return torch.device("cuda", _state["device"])


def _ensure_engine_built(engine_h: int, engine_w: int, first_x: Optional[torch.Tensor] = None) -> ops.Engine:
    """Build the native engine on the first frame, once its shape is known (reference depth.py:1842-1862).
    precision "fp8": the static activation scales are calibrated on that first batch of model inputs (the reference
    warms its engines up on the first frame the same way, depth.py:1861)."""
    if _state["cfg"] is None:
        raise _lib.D2SError("desktop2stereo_amd.depth.configure(...) has not been called")
    key = (engine_h, engine_w)
    if _state["engine"] is not None and _state["engine_key"] == key:
        return _state["engine"]
    with lock:
        if _state["engine"] is not None and _state["engine_key"] == key:
            return _state["engine"]
        if _state["engine"] is not None:
            _state["engine"].close()
        _state["engine"] = ops.Engine(_state["cfg"], _state["weights"], engine_h, engine_w, _state["max_batch"],
                                      _state["precision"], _state["device"], temporal=_state.get("temporal", False),
                                      max_depth=_state.get("max_depth", 0.0))
        if _state["precision"] in ("fp8", "fp8_mlp"):
            if first_x is None:
                raise _lib.D2SError("fp8 engine: no model inputs to calibrate on")
            _state["engine"].calibrate(first_x[: _state["max_batch"]])
        _state["engine_key"] = key
    return _state["engine"]


def calibrate(frames) -> None:
    """fp8 engines: (re)set the static e4m3 activation scales from REPRESENTATIVE frames (uint8 HWC / [B,H,W,3], numpy or
    tensor, all of one size; up to max_batch of them are used) instead of whatever frame happened to arrive first -- a black
    or splash first desktop frame gives ranges later frames saturate.  Builds the engine for that frame size if needed.
    D2S_FP8_HEADROOM (>= 1) widens every range.  No reference counterpart (the reference has FP16 only)."""
    if _state["precision"] not in ("fp8", "fp8_mlp"):
        raise _lib.D2SError("calibrate(): the engine is not configured with precision='fp8' / 'fp8_mlp'")
    t = torch.from_numpy(np.ascontiguousarray(frames)) if isinstance(frames, np.ndarray) else frames
    if t.dim() == 3:
        t = t.unsqueeze(0)
    t = t.to(device=_device())
    p = _state["params"]
    x = ops.preprocess(t, p.depth_resolution, _state["cfg"].patch, p.mean, p.std, p.resample, p.square_input)
    eng = _ensure_engine_built(int(x.shape[2]), int(x.shape[3]), x)
    eng.calibrate(x[: _state["max_batch"]])


def _fp8_first_inputs(frames_u8: torch.Tensor, key) -> Optional[torch.Tensor]:
    """Model inputs of the first batch, only when an fp8 engine is about to be built (calibration data)."""
    if _state["precision"] not in ("fp8", "fp8_mlp") or (_state["engine"] is not None and _state["engine_key"] == key):
        return None
    p = _state["params"]
    return ops.preprocess(frames_u8, p.depth_resolution, _state["cfg"].patch, p.mean, p.std, p.resample, p.square_input)


def process(img_uint8, target_height: int, branch: str = "cuda"):
    """The reference defines process() twice and picks one at import time by IS_CUDA (depth.py:540-629):

    branch="cuda" (default: what a CUDA *or ROCm* torch device runs, depth.py:540-566): HWC uint8 BGR / BGRA capture frame
      (numpy or tensor) -> CHW float32 RGB 0..255 on the device, down-scaled to `target_height` rows (even dims, bilinear +
      antialias) when the frame is taller.
    branch="cpu" (the other definition, depth.py:570-629), by input type like the reference:
      tensor -> its tensor branch (:576-601): already-RGB capture tensor [3|4,H,W] / [H,W,>=3], first three channels, plain
      bilinear (no antialias) to even dims, or the frame as is when target_height >= H0 -> device tensor CHW;
      numpy  -> its cv2 branch (:603-629): cvtColor BGR(A)->RGB + cv2.resize(INTER_AREA) to (int(W0*h/H0), h) -> uint8 HWC
      numpy array, like the reference returns (the UMat variant of the same branch is an OpenCL container: not built).
    The result feeds predict_depth / make_sbs."""
    if branch not in ("cuda", "cpu"):
        raise ValueError("process(): branch must be 'cuda' or 'cpu'")
    if branch == "cuda":
        t = torch.from_numpy(np.ascontiguousarray(img_uint8)) if isinstance(img_uint8, np.ndarray) else img_uint8
        return ops.process(t.to(device=_device()), target_height)
    if isinstance(img_uint8, torch.Tensor):
        return ops.process_rgb(img_uint8.to(device=_device()), target_height)
    t = torch.from_numpy(np.ascontiguousarray(img_uint8)).to(device=_device())
    return ops.process_area(t, target_height).cpu().numpy()


_FPS_MASK_CACHE = {"text": None, "frame": 0, "interval": 10}      # reference depth.py:2054-2059


def overlay_fps(rgb: torch.Tensor, fps: float) -> torch.Tensor:
    """Paint "FPS: %.1f" on a frame (reference depth.py:2061-2103).  Like the reference's cached mask, the
    text is rebuilt only every 10th call; unlike it, the frame is painted in place (callers pass a private copy)."""
    cache = _FPS_MASK_CACHE
    cache["frame"] += 1
    if cache["text"] is None or cache["frame"] % cache["interval"] == 0:
        cache["text"] = f"FPS: {fps:.1f}"
    return ops.overlay_text(rgb, cache["text"])


class DepthStabilizer:
    """EMA of the model-resolution depth across frames (reference depth.py:1865-1887); state on device."""

    def __init__(self, alpha=0.9):
        self.alpha = alpha
        self.prev = None
        self.enabled = True
        self.lock = Lock()

    def __call__(self, depth: torch.Tensor):
        if not self.enabled:
            return depth
        with self.lock:
            fresh = self.prev is None or self.prev.shape != depth.shape or self.prev.device != depth.device
            if fresh:
                self.prev = torch.empty_like(depth)
            ops.ema_update(depth, self.prev, not fresh, self.alpha)
            return depth if fresh else self.prev


depth_stabilizer = DepthStabilizer(alpha=0.9)


def post_process_depth(depth: torch.Tensor) -> torch.Tensor:
    """normalize -> gamma -> foreground scale -> anti-alias (reference depth.py:806-814)."""
    return ops.post_process_depth(depth, _state["params"])


def predict_depth(image_rgb, return_tuple=False, use_temporal_smooth: bool = True, dtype=None):
    """HWC uint8 RGB numpy frame or CHW tensor (0..255) -> [H,W] depth in [0,1] on the compute device, near ~ 1 (reference
    depth.py:1897-2025).  `dtype` (reference default: DTYPE = float16 with the FP16 setting, else float32; the reference only threads it
    through its XPU retry, depth.py:1959, its result carries the model's dtype): None / float32 return the float32 map the kernels
    produce, float16 / bfloat16 return it cast (what an FP16 reference hands its callers); anything else is a TypeError -- never
    silently ignored."""
    if dtype is not None and dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise TypeError(f"predict_depth(dtype=...) must be torch.float32, torch.float16 or torch.bfloat16, got {dtype!r}")
    p = _state["params"]
    if isinstance(image_rgb, torch.Tensor):
        rgb_tensor = image_rgb.to(device=_device())
        h, w = rgb_tensor.shape[1:]
        src = rgb_tensor if rgb_tensor.dtype in (torch.uint8, torch.float32) else rgb_tensor.float()
    else:
        h, w = image_rgb.shape[:2]
        hwc = torch.from_numpy(np.ascontiguousarray(image_rgb)).to(device=_device(), non_blocking=True)
        rgb_tensor = hwc.permute(2, 0, 1)
        src = hwc
    x = ops.preprocess(src, p.depth_resolution, _state["cfg"].patch if _state["cfg"] else 14, p.mean, p.std, p.resample, p.square_input)
    eng = _ensure_engine_built(x.shape[2], x.shape[3], x)
    depth = eng(x)
    depth = ops.post_process_depth(depth, p)[0]
    if use_temporal_smooth:
        depth = depth_stabilizer(depth)
    depth = ops.upsample_depth(depth, h, w)
    if dtype is not None and dtype != torch.float32:
        depth = depth.to(dtype)
    return (depth, rgb_tensor) if return_tuple else depth


def make_sbs_core(rgb: torch.Tensor, depth: torch.Tensor, ipd_uv=0.064, depth_ratio=2.0, display_mode="Half-SBS",
                  fill_16_9=False, convergence=0.0, device=None) -> torch.Tensor:
    """rgb [C,H,W] float (0..255), depth [H,W] -> [C,H',W'] float32 0..255 (reference depth.py:2122-2184)."""
    sp = ops.sbs_params(ipd_uv, depth_ratio, convergence, display_mode, fill_16_9)
    rgb = rgb.to(device=_device())
    if rgb.dtype != torch.uint8:
        rgb = rgb.float()
    return ops.make_sbs(rgb, depth.to(device=_device()), sp, _lib.FMT_F32_CHW)


def make_sbs(rgb_c, depth, ipd_uv=0.064, depth_ratio=2.0, convergence=0.0, fill_16_9=False, display_mode="Half-SBS", fps=None,
             inpaint=False):
    """-> HWC float32 numpy 0..255, like the reference (depth.py:2186-2231); `fps` paints the reference's
    FPS overlay on the source frame before the warp (depth.py:2216-2218).
    inpaint=True (north_star's to_stereo(..., inpaint=True)) renders the reference's OTHER warp instead: the GLSL DIBR
    shader with disocclusion in-painting of its Viewer modes (viewer.py:386-631), where the parallax is
    viewer.depth_strength (0.1) * depth_ratio in uv units and fill_16_9 is a window-layout matter that does not apply."""
    if inpaint:
        if fps is not None or fill_16_9:
            raise _lib.D2SError("make_sbs(inpaint=True): fps overlay / fill_16_9 belong to the torch warp (depth.py), "
                                "not to the viewer shader path")
        rgb = torch.from_numpy(np.ascontiguousarray(rgb_c)) if isinstance(rgb_c, np.ndarray) else rgb_c
        rgb = rgb.to(device=_device())
        if rgb.dim() == 3 and rgb.shape[0] == 3 and rgb.shape[-1] != 3:
            rgb = rgb.permute(1, 2, 0)
        if rgb.dtype != torch.uint8:                                   # viewer.py:2417: clamp(0,255).to(uint8)
            rgb = rgb.clamp(0, 255).to(torch.uint8)
        d = torch.from_numpy(depth) if isinstance(depth, np.ndarray) else depth
        dp = ops.dibr_params(ipd_uv, depth_ratio, convergence, display_mode)
        return ops.dibr_warp(rgb.contiguous(), d.to(device=_device()), dp, out_u8=False).cpu().numpy()
    if isinstance(depth, np.ndarray):
        depth = torch.from_numpy(depth)
    depth = depth.to(device=_device())
    if isinstance(rgb_c, np.ndarray):
        rgb = torch.from_numpy(np.ascontiguousarray(rgb_c)).to(device=_device())       # HWC uint8 / float
        if rgb.dtype != torch.uint8:
            rgb = rgb.float()
            if rgb.dim() == 3 and rgb.shape[2] == 3:              # HWC -> CHW only when it IS HWC (reference depth.py:2205-2207)
                rgb = rgb.permute(2, 0, 1).contiguous()
    else:
        rgb = rgb_c.to(device=_device())
        if rgb.dtype != torch.uint8:
            rgb = rgb.float()
    if fps is not None:
        rgb = rgb.contiguous()
        if isinstance(rgb_c, torch.Tensor) and rgb.data_ptr() == rgb_c.data_ptr():
            rgb = rgb.clone()                                   # never paint the caller's frame
        rgb = overlay_fps(rgb, fps)
    sp = ops.sbs_params(ipd_uv, depth_ratio, convergence, display_mode, fill_16_9)
    return ops.make_sbs(rgb, depth, sp, _lib.FMT_F32_HWC).cpu().numpy()


def pipeline(frames, display_mode=None, use_temporal_smooth=False, out_u8=True, want_depth=False):
    """Batched predict_depth + make_sbs: uint8 [B,H,W,3] (numpy or device tensor) -> device tensor
    [B,H',W',3] (uint8, or float32 when out_u8=False) in one stream-ordered native call."""
    p = _state["params"]
    t = torch.from_numpy(np.ascontiguousarray(frames)) if isinstance(frames, np.ndarray) else frames
    t = t.to(device=_device())
    B, H, W, _ = t.shape
    h, w, _s = engine_shape(H, W, p.depth_resolution, _state["cfg"].patch, p.square_input)
    if B > _state["max_batch"]:
        raise _lib.D2SError(f"batch {B} > configured max_batch {_state['max_batch']}")
    eng = _ensure_engine_built(h, w, _fp8_first_inputs(t, (h, w)))
    sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, display_mode or p.display_mode, p.fill_16_9)
    return eng.pipeline(t, p, sp, use_ema=use_temporal_smooth, out_fmt=_lib.FMT_U8_HWC if out_u8 else _lib.FMT_F32_HWC,
                        want_depth=want_depth)


def pipeline_mixed(frames_list, display_mode=None, out_u8=True):
    """Mixed-resolution batch (BASELINE config 5): a list of uint8 HWC frames of different sizes whose
    model-input shape is the same (every 16:9 frame maps to 294x518 at Depth Resolution 518, reference
    depth.py:676-706).  Pre-process per frame size, ONE batched model + post-process pass over all frames,
    then the warp per frame size.  Returns a list of device tensors in input order."""
    p = _state["params"]
    cfg = _state["cfg"]
    dev = _device()
    ts = [torch.from_numpy(np.ascontiguousarray(f)).to(dev) if isinstance(f, np.ndarray) else f.to(dev) for f in frames_list]
    shapes = {engine_shape(t.shape[0], t.shape[1], p.depth_resolution, cfg.patch, p.square_input)[:2] for t in ts}
    if len(shapes) != 1:
        raise _lib.D2SError(f"pipeline_mixed: frames map to different model-input shapes {sorted(shapes)}")
    h, w = shapes.pop()
    if len(ts) > _state["max_batch"]:
        raise _lib.D2SError(f"batch {len(ts)} > configured max_batch {_state['max_batch']}")
    eng = _ensure_engine_built(h, w, _fp8_first_inputs(ts[0].unsqueeze(0), (h, w)))
    groups = {}
    for i, t in enumerate(ts):
        groups.setdefault(tuple(t.shape[:2]), []).append(i)
    x = torch.empty((len(ts), 3, h, w), dtype=torch.float32, device=dev)
    for (H, W), idx in groups.items():
        x[idx] = ops.preprocess(torch.stack([ts[i] for i in idx]), p.depth_resolution, cfg.patch, p.mean, p.std, p.resample, p.square_input)
    depth = ops.post_process_depth(eng(x), p)
    sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, display_mode or p.display_mode, p.fill_16_9)
    out = [None] * len(ts)
    for (H, W), idx in groups.items():
        o = ops.make_sbs(torch.stack([ts[i] for i in idx]), depth[idx], sp, _lib.FMT_U8_HWC if out_u8 else _lib.FMT_F32_HWC)
        for j, i in enumerate(idx):
            out[i] = o[j]
    return out


# names used by BASELINE.json's north_star
predict = predict_depth
to_stereo = make_sbs
