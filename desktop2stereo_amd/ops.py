"""Stage-level host wrappers over the C-ABI (include/d2s.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every computation
is a call into libd2s_hip.so with ``tensor.data_ptr()`` (the same convention as the reference's
MIGraphXEngine.__call__, reference depth.py:1029-1045).  Nothing in this module computes with
torch ops, and nothing falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import (FMT_F32_CHW, FMT_F32_HWC, FMT_U8_CHW, FMT_U8_HWC, MODE, PREC_BF16, PREC_FP32,
                   ModelDesc, PostParams, PreParams, SbsParams, check)
from .config import IMAGENET_MEAN, IMAGENET_STD, ModelConfig, PipelineParams, engine_shape


class _on:
    """Scope of one native call on `device`: the HIP current device is per host thread and the reference issues
    predict_depth and make_sbs from different threads, so the device of the tensors is made
    current around the call and the stream handed to the library is THAT device's current stream (not the calling
    thread's default device's).  `with _on(t.device) as st: lib.d2s_...(..., st)`."""

    def __init__(self, device: torch.device):
        self.device = device
        self.guard = torch.cuda.device(device)

    def __enter__(self) -> C.c_void_p:
        self.guard.__enter__()
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __exit__(self, *exc):
        return self.guard.__exit__(*exc)


def _same_device(a: torch.Tensor, b: torch.Tensor, what: str):
    if a.device != b.device:
        raise _lib.D2SError(f"{what}: tensors on different devices ({a.device} vs {b.device})")


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.D2SError(f"{what} must be a ROCm device tensor (no CPU path in this package)")


def post_params(p: PipelineParams) -> PostParams:
    return PostParams(p.percentile, p.subsample_cap, p.gamma, p.foreground_scale, p.aa_strength, p.ema_alpha,
                      int(bool(p.metric)))


def pre_params(mean=IMAGENET_MEAN, std=IMAGENET_STD, resample: str = "bilinear", square: bool = False) -> PreParams:
    if resample not in _lib.RESAMPLE:
        raise ValueError(f"resample must be one of {list(_lib.RESAMPLE)}")
    return PreParams((C.c_float * 3)(*mean), (C.c_float * 3)(*std), _lib.RESAMPLE[resample], int(bool(square)))


def sbs_params(ipd_uv=0.064, depth_ratio=2.0, convergence=0.0, display_mode="Half-SBS", fill_16_9=False) -> SbsParams:
    if display_mode not in MODE:
        raise ValueError(f"display_mode must be one of {list(MODE)}")
    return SbsParams(float(ipd_uv), float(depth_ratio), float(convergence), MODE[display_mode], int(bool(fill_16_9)))


def sbs_shape(H: int, W: int, sp: SbsParams) -> Tuple[int, int]:
    oh, ow = C.c_int(), C.c_int()
    check(_lib.load().d2s_sbs_shape(H, W, C.byref(sp), C.byref(oh), C.byref(ow)), "d2s_sbs_shape")
    return oh.value, ow.value


def _frame_fmt(t: torch.Tensor) -> Tuple[int, int, int, int]:
    """(fmt, batch, H, W) of a frame tensor: u8 HWC / u8 CHW / f32 CHW, optional leading batch."""
    if t.dtype == torch.uint8 and t.shape[-1] == 3 and t.dim() in (3, 4):
        b = t.shape[0] if t.dim() == 4 else 1
        return FMT_U8_HWC, b, t.shape[-3], t.shape[-2]
    if t.shape[-3] == 3 and t.dim() in (3, 4):
        b = t.shape[0] if t.dim() == 4 else 1
        if t.dtype == torch.uint8:
            return FMT_U8_CHW, b, t.shape[-2], t.shape[-1]
        if t.dtype == torch.float32:
            return FMT_F32_CHW, b, t.shape[-2], t.shape[-1]
    raise ValueError(f"unsupported frame tensor {tuple(t.shape)} {t.dtype}: want uint8 [..,H,W,3], uint8/float32 [..,3,H,W]")


def process(img_bgr: torch.Tensor, target_height: int) -> torch.Tensor:
    """A1 (reference depth.py:540-566): uint8 HWC BGR(A) device tensor -> float32 CHW RGB 0..255, anti-aliased
    bilinear down-scale to even dims when target_height < H0."""
    _need_cuda(img_bgr, "img")
    if img_bgr.dtype != torch.uint8 or img_bgr.dim() != 3 or img_bgr.shape[-1] not in (3, 4):
        raise ValueError(f"process(): want uint8 [H,W,3|4] (BGR / BGRA), got {tuple(img_bgr.shape)} {img_bgr.dtype}")
    img_bgr = img_bgr.contiguous()
    H0, W0, ch = img_bgr.shape
    lib = _lib.load()
    oh, ow = C.c_int(), C.c_int()
    check(lib.d2s_process_shape(H0, W0, int(target_height), C.byref(oh), C.byref(ow)), "d2s_process_shape")
    out = torch.empty((3, oh.value, ow.value), dtype=torch.float32, device=img_bgr.device)
    with _on(img_bgr.device) as st:
        check(lib.d2s_process(_ptr(img_bgr), ch, H0, W0, int(target_height), _ptr(out), st), "d2s_process")
    return out


def process_rgb(img: torch.Tensor, target_height: int) -> torch.Tensor:
    """A1, tensor branch of the reference's non-CUDA process() (depth.py:576-601): RGB capture tensor [3|4,H,W] or [H,W,>=3]
    (uint8 or float32) -> first three channels as CHW; target_height < H0: float32 bilinear (no antialias) down-scale to even
    dims; else the frame itself (dtype unchanged, like the reference)."""
    _need_cuda(img, "img")
    if img.dim() != 3:
        raise ValueError(f"Unsupported tensor image shape: {tuple(img.shape)}")
    if img.shape[0] in (3, 4):
        chw, ch, H0, W0 = True, img.shape[0], img.shape[1], img.shape[2]
    elif img.shape[-1] >= 3:
        chw, ch, H0, W0 = False, img.shape[2], img.shape[0], img.shape[1]
    else:
        raise ValueError(f"Unsupported tensor image shape: {tuple(img.shape)}")
    if target_height >= H0:                                       # depth.py:586-587: returned as is (no resize, no cast)
        return img[:3] if chw else img[..., :3].permute(2, 0, 1).contiguous()
    if img.dtype not in (torch.uint8, torch.float32):
        img = img.float()
    if not chw and (img.dtype != torch.uint8 or ch > 4):
        img, chw, ch = img[..., :3].permute(2, 0, 1), True, 3      # float HWC / wide HWC: plane view first
    img = img.contiguous()
    fmt = (FMT_U8_CHW if img.dtype == torch.uint8 else FMT_F32_CHW) if chw else FMT_U8_HWC
    lib = _lib.load()
    oh, ow = C.c_int(), C.c_int()
    check(lib.d2s_process_shape(H0, W0, int(target_height), C.byref(oh), C.byref(ow)), "d2s_process_shape")
    out = torch.empty((3, oh.value, ow.value), dtype=torch.float32, device=img.device)
    with _on(img.device) as st:
        check(lib.d2s_process_rgb(_ptr(img), fmt, ch, H0, W0, int(target_height), _ptr(out), st), "d2s_process_rgb")
    return out


def process_area(img_bgr: torch.Tensor, target_height: int) -> torch.Tensor:
    """A1, numpy branch of the reference's non-CUDA process() (depth.py:603-629): uint8 HWC BGR(A) device tensor -> uint8 HWC RGB,
    cv2.resize(INTER_AREA) to (int(W0*height/H0), height) when height < H0."""
    _need_cuda(img_bgr, "img")
    if img_bgr.dtype != torch.uint8 or img_bgr.dim() != 3 or img_bgr.shape[-1] not in (3, 4):
        raise ValueError(f"process(): want uint8 [H,W,3|4] (BGR / BGRA), got {tuple(img_bgr.shape)} {img_bgr.dtype}")
    img_bgr = img_bgr.contiguous()
    H0, W0, ch = img_bgr.shape
    lib = _lib.load()
    oh, ow = C.c_int(), C.c_int()
    check(lib.d2s_process_area_shape(H0, W0, int(target_height), C.byref(oh), C.byref(ow)), "d2s_process_area_shape")
    out = torch.empty((oh.value, ow.value, 3), dtype=torch.uint8, device=img_bgr.device)
    with _on(img_bgr.device) as st:
        check(lib.d2s_process_area(_ptr(img_bgr), ch, H0, W0, int(target_height), _ptr(out), st), "d2s_process_area")
    return out


def overlay_text(frame: torch.Tensor, text: str) -> torch.Tensor:
    """A15 (reference depth.py:2061-2103): paint `text` in the reference's 5x3 font, green, IN PLACE on one frame."""
    _need_cuda(frame, "frame")
    if not frame.is_contiguous():
        raise ValueError("overlay_text paints in place: the frame must be contiguous")
    if frame.dtype == torch.float32 and frame.dim() == 3 and frame.shape[-1] == 3 and frame.shape[0] != 3:
        fmt, H, W = FMT_F32_HWC, frame.shape[0], frame.shape[1]
    else:
        fmt, B, H, W = _frame_fmt(frame)
        if B != 1:
            raise ValueError("overlay_text takes one frame")
    with _on(frame.device) as st:
        check(_lib.load().d2s_overlay_text(_ptr(frame), fmt, H, W, text.encode(), st), "d2s_overlay_text")
    return frame


def preprocess(frames: torch.Tensor, target: int, patch: int = 14, mean=IMAGENET_MEAN, std=IMAGENET_STD,
               resample: str = "bilinear", square: bool = False) -> torch.Tensor:
    """A2-A4 (reference depth.py:676-706, 1916-1948) -> float32 [B,3,h,w].  resample: "bilinear" = the CPU branch of
    _resize_patch_aligned_t (decimation + bilinear), "bicubic_aa" = its IS_CUDA branch (depth.py:698-699).
    square=True: the fixed-square branch (get_patch_size() is None, CAPTURE_MODE "Window"; depth.py:1937-1946): plain bilinear
    of the full frame to target x target."""
    _need_cuda(frames, "frames")
    frames = frames.contiguous()
    fmt, B, H, W = _frame_fmt(frames)
    h, w, stride = engine_shape(H, W, target, patch, square)
    out = torch.empty((B, 3, h, w), dtype=torch.float32, device=frames.device)
    pre = pre_params(mean, std, resample, square)
    with _on(frames.device) as st:
        check(_lib.load().d2s_preprocess(_ptr(frames), fmt, B, H, W, _ptr(out), h, w, stride, C.byref(pre), st), "d2s_preprocess")
    return out


def post_process_depth(depth: torch.Tensor, p: PipelineParams) -> torch.Tensor:
    """A10-A11 (reference depth.py:806-814) on float32 [B,h,w] or [h,w]; returns a new tensor."""
    _need_cuda(depth, "depth")
    d = depth.to(torch.float32).contiguous().clone()
    B = d.shape[0] if d.dim() == 3 else 1
    h, w = d.shape[-2:]
    lib = _lib.load()
    nbytes = lib.d2s_post_process_workspace(B, h, w)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d.device)
    pp = post_params(p)
    with _on(d.device) as st:
        check(lib.d2s_post_process(_ptr(d), B, h, w, C.byref(pp), _ptr(ws), nbytes, st), "d2s_post_process")
    return d


def post_process_depth_to(depth: torch.Tensor, p: PipelineParams) -> torch.Tensor:
    """The same through d2s_post_process_to (out of place: with few frames the one-launch form runs); the input is left untouched."""
    _need_cuda(depth, "depth")
    d = depth.to(torch.float32).contiguous()
    out = torch.empty_like(d)
    B = d.shape[0] if d.dim() == 3 else 1
    h, w = d.shape[-2:]
    lib = _lib.load()
    nbytes = lib.d2s_post_process_workspace(B, h, w)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d.device)
    pp = post_params(p)
    with _on(d.device) as st:
        check(lib.d2s_post_process_to(_ptr(d), _ptr(out), B, h, w, C.byref(pp), _ptr(ws), nbytes, st), "d2s_post_process_to")
    return out


def ema_update(depth: torch.Tensor, state: torch.Tensor, initialised: bool, alpha: float) -> torch.Tensor:
    """A12 (reference depth.py:1865-1887); depth [h,w] is overwritten with the returned value."""
    h, w = depth.shape
    _same_device(depth, state, "ema_update")
    with _on(depth.device) as st:
        check(_lib.load().d2s_ema_update(_ptr(depth), _ptr(state), int(initialised), h, w, alpha, st), "d2s_ema_update")
    return depth


def upsample_depth(depth: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """A13 (reference depth.py:1999-2004): [B,h,w] or [h,w] -> same rank at H x W."""
    _need_cuda(depth, "depth")
    d = depth.to(torch.float32).contiguous()
    B = d.shape[0] if d.dim() == 3 else 1
    h, w = d.shape[-2:]
    out = torch.empty((B, H, W) if d.dim() == 3 else (H, W), dtype=torch.float32, device=d.device)
    with _on(d.device) as st:
        check(_lib.load().d2s_upsample_depth(_ptr(d), B, h, w, _ptr(out), H, W, st), "d2s_upsample_depth")
    return out


def make_sbs(frames: torch.Tensor, depth: torch.Tensor, sp: SbsParams, out_fmt: int = FMT_U8_HWC, out: torch.Tensor = None) -> torch.Tensor:
    """A14 (+A13 fused when depth is at model resolution) (reference depth.py:2122-2184).  `out`: a caller-allocated result (the
    C-ABI's own convention), shape [B, oh, ow, 3] / [B, 3, oh, ow] of the format's dtype."""
    _need_cuda(frames, "frames")
    _need_cuda(depth, "depth")
    _same_device(frames, depth, "make_sbs")
    frames = frames.contiguous()
    d = depth.to(torch.float32).contiguous()
    fmt, B, H, W = _frame_fmt(frames)
    dB = d.shape[0] if d.dim() == 3 else 1
    if dB != B:
        raise ValueError("frames / depth batch mismatch")
    dh, dw = d.shape[-2:]
    oh, ow = sbs_shape(H, W, sp)
    batched = frames.dim() == 4
    if out is not None:
        want = ((B, 3, oh, ow) if out_fmt == FMT_F32_CHW else (B, oh, ow, 3), torch.uint8 if out_fmt == FMT_U8_HWC else torch.float32)
        if tuple(out.shape) != want[0] or out.dtype != want[1] or not out.is_contiguous() or out.device != frames.device:
            raise ValueError(f"make_sbs: out must be a contiguous {want[1]} tensor of shape {want[0]} on {frames.device}")
    elif out_fmt == FMT_U8_HWC:
        out = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=frames.device)
    elif out_fmt == FMT_F32_HWC:
        out = torch.empty((B, oh, ow, 3), dtype=torch.float32, device=frames.device)
    elif out_fmt == FMT_F32_CHW:
        out = torch.empty((B, 3, oh, ow), dtype=torch.float32, device=frames.device)
    else:
        raise ValueError("bad out_fmt")
    with _on(frames.device) as st:
        check(_lib.load().d2s_make_sbs(_ptr(frames), fmt, _ptr(d), dh, dw, B, H, W, C.byref(sp), _ptr(out), out_fmt, st),
              "d2s_make_sbs")
    return out if batched else out[0]


def dibr_params(ipd_uv=0.064, depth_ratio=1.0, convergence=0.0, display_mode="Full-SBS", roll=0.0, feather=False,
                viewer_depth_strength=0.1, search_radius=12.0, depth_tolerance=0.012, blur_radius=2.5,
                feather_width=0.02, resolution=(0.0, 0.0), corner_radius=0.0, viewport=(0.0, 0.0, 0.0, 0.0),
                alpha="window") -> _lib.DibrParams:
    """Uniform block of the reference's DIBR shader with the viewer's defaults (viewer.py:1333-1343, 402-411).
    corner_radius: u_corner_radius (0 desktop viewer, 0.03 OpenXR screen); viewport: u_viewport (x, y, w, h) in pixels of
    the eye image, y up -- zeros = the eye image itself.  alpha: "window" = frag_color.rgb as the reference's window shows it
    (its stereo quads are drawn with blending off), "premultiplied" = rgb * a, "rgba" = four channels (include/d2s.h)."""
    if display_mode not in MODE:
        raise ValueError(f"display_mode must be one of {list(MODE)}")
    if alpha not in _lib.DIBR_ALPHA:
        raise ValueError(f"alpha must be one of {list(_lib.DIBR_ALPHA)}")
    return _lib.DibrParams(float(ipd_uv), float(viewer_depth_strength * depth_ratio), float(convergence), float(roll),
                           float(search_radius), float(depth_tolerance), float(blur_radius), float(resolution[0]),
                           float(resolution[1]), MODE[display_mode], int(bool(feather)), float(feather_width),
                           float(corner_radius), (C.c_float * 4)(*[float(v) for v in viewport]), _lib.DIBR_ALPHA[alpha],
                           C.sizeof(_lib.DibrParams))


def dibr_warp(frames: torch.Tensor, depth: torch.Tensor, dp: "_lib.DibrParams", out_u8: bool = True) -> torch.Tensor:
    """f1 (reference viewer.py:386-631): uint8 HWC frames [B,H,W,3] or [H,W,3] + full-resolution depth -> both eyes
    with disocclusion in-painting, packed per dp.display_mode."""
    _need_cuda(frames, "frames")
    _need_cuda(depth, "depth")
    if frames.dtype != torch.uint8 or frames.shape[-1] != 3 or frames.dim() not in (3, 4):
        raise ValueError("dibr_warp: frames must be uint8 [B,H,W,3] or [H,W,3]")
    batched = frames.dim() == 4
    f = frames.contiguous() if batched else frames.contiguous().unsqueeze(0)
    d = depth.to(torch.float32).contiguous()
    d = d if d.dim() == 3 else d.unsqueeze(0)
    B, H, W, _ = f.shape
    if tuple(d.shape) != (B, H, W):
        raise ValueError(f"dibr_warp: depth must be full resolution {(B, H, W)}, got {tuple(d.shape)}")
    lib = _lib.load()
    oh, ow = C.c_int(), C.c_int()
    check(lib.d2s_dibr_shape(H, W, dp.display_mode, C.byref(oh), C.byref(ow)), "d2s_dibr_shape")
    nch = 4 if dp.alpha_mode == _lib.DIBR_ALPHA["rgba"] else 3
    out = torch.empty((B, oh.value, ow.value, nch), dtype=torch.uint8 if out_u8 else torch.float32, device=f.device)
    _same_device(f, d, "dibr_warp")
    with _on(f.device) as st:
        check(lib.d2s_dibr_warp(_ptr(f), _ptr(d), B, H, W, C.byref(dp), _ptr(out), FMT_U8_HWC if out_u8 else FMT_F32_HWC, st),
              "d2s_dibr_warp")
    return out if batched else out[0]


_JPEG_WS: Dict[Tuple[int, int], torch.Tensor] = {}


def jpeg_bound(H: int, W: int) -> Tuple[int, int]:
    """(output bytes that can never overflow, workspace bytes per frame) for an H x W frame."""
    ob, wb = C.c_int64(), C.c_int64()
    check(_lib.load().d2s_jpeg_bound(H, W, C.byref(ob), C.byref(wb)), "d2s_jpeg_bound")
    return ob.value, wb.value


def jpeg_encode(frames: torch.Tensor, quality: int = 90, out_stride: Optional[int] = None,
                workspace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """f3, the MJPEG sink (reference streamer.py:249-256, 285-291 — cv2.imencode('.jpg', ...)): RGB frames
    [B,H,W,3] or [H,W,3], uint8 or float32 0..255 (rounded half-even + saturated like cv2's convertTo), on the
    device -> (bytes [B, out_stride] uint8, sizes [B] int32), both on the device, no host sync.  The first
    sizes[b] bytes of row b are the JPEG libjpeg-turbo would write for that frame at that quality (4:2:0).
    workspace: uint8 device scratch of >= B * jpeg_bound(H, W)[1] + 256 bytes owned by the caller (one per stream that
    encodes concurrently); default: a module-level buffer, good for one stream at a time."""
    _need_cuda(frames, "frames")
    if frames.shape[-1] != 3 or frames.dim() not in (3, 4) or frames.dtype not in (torch.uint8, torch.float32):
        raise ValueError("jpeg_encode: frames must be uint8/float32 [B,H,W,3] or [H,W,3]")
    f = (frames if frames.dim() == 4 else frames.unsqueeze(0)).contiguous()
    B, H, W, _ = f.shape
    safe, ws_frame = jpeg_bound(H, W)
    # default stride: the unstuffed worst case + 1/16 for 0xFF stuffing; sizes[b] = -1 reports an overflow
    stride = int(out_stride) if out_stride else (safe // 2 + safe // 32 + 1024)
    if workspace is not None:
        if workspace.dtype != torch.uint8 or not workspace.is_cuda or workspace.numel() < ws_frame * B + 256:
            raise ValueError(f"jpeg_encode: workspace must be a uint8 device tensor of >= {ws_frame * B + 256} bytes")
        ws = workspace
    else:
        key = (f.device.index or 0, torch.cuda.current_stream(f.device).cuda_stream)
        ws = _JPEG_WS.get(key)                      # one growing scratch per (device, stream): never freed under a running encode
        if ws is None or ws.numel() < ws_frame * B + 256:
            ws = _JPEG_WS[key] = torch.empty(ws_frame * B + 256, dtype=torch.uint8, device=f.device)
    pad = (-ws.data_ptr()) % 256
    out = torch.empty((B, stride), dtype=torch.uint8, device=f.device)
    sizes = torch.empty((B,), dtype=torch.int32, device=f.device)
    _same_device(f, ws, "jpeg_encode workspace")
    with _on(f.device) as st:
        check(_lib.load().d2s_jpeg_encode(_ptr(f), FMT_U8_HWC if f.dtype == torch.uint8 else FMT_F32_HWC, B, H, W, int(quality),
                                          _ptr(out), stride, _ptr(sizes), C.c_void_p(ws.data_ptr() + pad), ws_frame * B, st),
              "d2s_jpeg_encode")
    return out, sizes


class Engine:
    """The native depth engine: what DepthModelWrapper holds in ``self.model`` for an accelerated
    backend (reference depth.py:1539-1781).  ``__call__(tensor[B,3,h,w]) -> tensor[B,h,w]``."""

    def __init__(self, cfg: ModelConfig, weights: Dict[str, np.ndarray], h: int, w: int, max_batch: int = 1,
                 precision: str = "bf16", device: int = 0, temporal: bool = False, max_depth: float = 0.0):
        """temporal=True: streaming Video-Depth-Anything (weights from vda_weights; one stream, batch 1).
        max_depth > 0: metric head, sigmoid * max_depth (HF depth_estimation_type "metric")."""
        if not torch.cuda.is_available():
            raise _lib.D2SError("no ROCm device: the HIP engine cannot run (and there is no fallback)")
        self.lib = _lib.load()
        self.cfg, self.h, self.w, self.max_batch = cfg, h, w, max_batch
        self.precision = precision
        self.device = torch.device("cuda", device)
        desc = ModelDesc(cfg.hidden, cfg.heads, cfg.layers, (C.c_int32 * 4)(*cfg.out_indices), (C.c_int32 * 4)(*cfg.neck),
                         cfg.fusion, cfg.head_hidden, cfg.mlp, cfg.patch, cfg.pos_grid, cfg.ln_eps,
                         {"bf16": PREC_BF16, "fp32": PREC_FP32, "fp8": _lib.PREC_FP8, "bf16x3": _lib.PREC_BF16X3, "fp8_mlp": _lib.PREC_FP8_MLP}.get(precision, -1),
                         int(bool(temporal)), float(max_depth))
        if precision not in ("bf16", "fp32", "fp8", "bf16x3", "fp8_mlp"):
            raise ValueError("precision must be 'bf16', 'fp32', 'fp8', 'fp8_mlp' or 'bf16x3'")
        self._h = C.c_void_p()
        with _on(self.device):
            check(self.lib.d2s_engine_create(C.byref(desc), device, C.byref(self._h)), "d2s_engine_create")
            for name, arr in weights.items():
                a = np.ascontiguousarray(arr, dtype=np.float32)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                check(self.lib.d2s_engine_set_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                      f"d2s_engine_set_weight({name})")
            check(self.lib.d2s_engine_finalize(self._h, h, w, max_batch), "d2s_engine_finalize")

    def _mine(self, t: torch.Tensor, what: str):
        _need_cuda(t, what)
        if t.device != self.device:
            raise _lib.D2SError(f"{what} is on {t.device}, the engine lives on {self.device}")

    def memory_bytes(self) -> int:
        b = C.c_uint64()
        check(self.lib.d2s_engine_memory(self._h, C.byref(b)))
        return b.value

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        self._mine(x, "pixel_values")
        x = x.to(torch.float32).contiguous()
        if x.dim() == 3:
            x = x.unsqueeze(0)
        B = x.shape[0]
        if tuple(x.shape[1:]) != (3, self.h, self.w):
            raise ValueError(f"engine was built for [B,3,{self.h},{self.w}], got {tuple(x.shape)}")
        out = torch.empty((B, self.h, self.w), dtype=torch.float32, device=x.device)
        with _on(self.device) as st:
            check(self.lib.d2s_model_forward(self._h, _ptr(x), _ptr(out), B, st), "d2s_model_forward")
        return out

    def calibrate(self, x: torch.Tensor):
        """fp8 engines: set the static activation scales from one bf16 pass over calibration inputs x [B,3,h,w]
        (normalised model inputs, e.g. ops.preprocess of representative frames).  Required before the first forward."""
        self._mine(x, "calibration inputs")
        x = x.to(torch.float32).contiguous()
        if x.dim() == 3:
            x = x.unsqueeze(0)
        if tuple(x.shape[1:]) != (3, self.h, self.w):
            raise ValueError(f"engine was built for [B,3,{self.h},{self.w}], got {tuple(x.shape)}")
        with _on(self.device) as st:
            check(self.lib.d2s_engine_calibrate(self._h, _ptr(x), x.shape[0], st), "d2s_engine_calibrate")

    def tap(self, name: str) -> torch.Tensor:
        rows, cols = C.c_int(), C.c_int()
        n = max(self.cfg.hidden * (self.h // 14 * (self.w // 14) + 1), 16 * (self.h // 14) * (self.w // 14) * self.cfg.fusion)
        buf = torch.empty(n, dtype=torch.float32, device=self.device)
        with _on(self.device) as st:
            check(self.lib.d2s_engine_tap(self._h, name.encode(), _ptr(buf), n, C.byref(rows), C.byref(cols), st), "d2s_engine_tap")
        return buf[: rows.value * cols.value].view(rows.value, cols.value)

    def profile(self, enable: bool):
        """Start (and clear) / stop HIP-event timing around every kernel launch."""
        check(self.lib.d2s_engine_profile(self._h, int(enable)), "d2s_engine_profile")

    def profile_read(self) -> Dict[str, dict]:
        n = 16
        ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        nc = C.c_int()
        check(self.lib.d2s_engine_profile_read(self._h, n, ms, fl, by, cnt, C.byref(nc)), "d2s_engine_profile_read")
        return {self.lib.d2s_profile_class_name(i).decode(): {"ms": ms[i], "flops": fl[i], "bytes": by[i], "launches": cnt[i]}
                for i in range(nc.value)}

    def reset_stream(self):
        check(self.lib.d2s_engine_reset_stream(self._h))

    def pipeline(self, frames: torch.Tensor, p: PipelineParams, sp: SbsParams, use_ema: bool = False,
                 out_fmt: int = FMT_U8_HWC, want_depth: bool = False, out: Optional[torch.Tensor] = None):
        """predict_depth + make_sbs for uint8 HWC frames [B,H,W,3] in one stream-ordered call."""
        self._mine(frames, "frames")
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("frames must be uint8 [B,H,W,3]")
        frames = frames.contiguous()
        B, H, W, _ = frames.shape
        oh, ow = sbs_shape(H, W, sp)
        if out is None:
            if out_fmt == FMT_U8_HWC:
                out = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=frames.device)
            elif out_fmt == FMT_F32_HWC:
                out = torch.empty((B, oh, ow, 3), dtype=torch.float32, device=frames.device)
            else:
                out = torch.empty((B, 3, oh, ow), dtype=torch.float32, device=frames.device)
        else:
            self._mine(out, "out")
        depth = torch.empty((B, H, W), dtype=torch.float32, device=frames.device) if want_depth else None
        pp = post_params(p)
        pre = pre_params(p.mean, p.std, p.resample, p.square_input)
        with _on(self.device) as st:
            check(self.lib.d2s_pipeline(self._h, _ptr(frames), B, H, W, p.depth_resolution, C.byref(pre), C.byref(pp), C.byref(sp),
                                        int(use_ema), _ptr(out), out_fmt, _ptr(depth) if want_depth else None, st), "d2s_pipeline")
        return (out, depth) if want_depth else out

    def close(self):
        if getattr(self, "_h", None):
            self.lib.d2s_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reload_env() -> int:
    """Make libd2s_hip.so re-read its kernel-selection switches (D2S_NO_HALO2, D2S_NO_WIDE, ...) from the environment."""
    return int(_lib.load().d2s_debug_reload_env())


def gemm_probe(A: torch.Tensor, Wt: torch.Tensor, bias: Optional[torch.Tensor], precision: str, tile: int = 0, iters: int = 1):
    """C = A @ Wt^T (+bias) through the engine's MFMA kernel (test / micro-benchmark)."""
    M, K = A.shape
    N = Wt.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    check(_lib.load().d2s_gemm_probe(_ptr(A.contiguous()), _ptr(Wt.contiguous()), _ptr(bias) if bias is not None else None,
                                     _ptr(out), M, N, K, {"bf16": PREC_BF16, "fp32": PREC_FP32, "fp8": _lib.PREC_FP8, "bf16x3": _lib.PREC_BF16X3}[precision], tile, iters,
                                     C.c_void_p(torch.cuda.current_stream(A.device).cuda_stream)),
          "d2s_gemm_probe")
    return out


def attention_probe(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, precision: str = "bf16", iters: int = 1):
    """softmax(q k^T / 8) v for float32 [B, heads, N, 64] device tensors through the engine's attention kernel
    (test / micro-benchmark).  Returns (out [B, N, heads * 64] float32, ms per launch or 0)."""
    B, H, N, d = q.shape
    if d != 64 or k.shape != q.shape or v.shape != q.shape:
        raise ValueError("attention_probe: q, k, v must be [B, heads, N, 64]")
    out = torch.empty((B, N, H * 64), dtype=torch.float32, device=q.device)
    ms = C.c_float(0.0)
    with _on(q.device) as st:
        check(_lib.load().d2s_attention_probe(_ptr(q.float().contiguous()), _ptr(k.float().contiguous()), _ptr(v.float().contiguous()),
                                              _ptr(out), B, H, N, {"bf16": PREC_BF16, "fp32": PREC_FP32, "bf16x3": _lib.PREC_BF16X3}[precision], iters,
                                              C.byref(ms), st), "d2s_attention_probe")
    return out, ms.value
