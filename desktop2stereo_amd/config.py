"""Hot-path parameters of the 2D->3D pipeline.

The reference reads ~10 module-level constants out of ``settings.yaml`` through
``utils.py`` (reference utils.py:834-859, 900; defaults settings.yaml:315-323, 350).
This module keeps exactly those, as plain dataclasses, and the Depth-Anything-v2
architecture table (HF ``config.json`` of depth-anything/Depth-Anything-V2-*-hf;
dims restated in SURVEY.md section 8).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Tuple

PATCH = 14                       # DINOv2 patch size (reference depth.py:531-538)
POS_GRID = 37                    # 518 / 14: pre-trained position-embedding grid
IMAGENET_MEAN = (0.485, 0.456, 0.406)   # reference depth.py:1798
IMAGENET_STD = (0.229, 0.224, 0.225)    # reference depth.py:1799

DISPLAY_MODES = ("Half-SBS", "Full-SBS", "Half-TAB", "Full-TAB")  # depth.py:2178-2183


@dataclass(frozen=True)
class ModelConfig:
    """Depth-Anything-v2 (DINOv2 backbone + DPT neck/head) dimensions."""
    name: str
    hidden: int
    heads: int
    layers: int
    out_indices: Tuple[int, int, int, int]
    neck: Tuple[int, int, int, int]
    fusion: int
    head_hidden: int = 32
    mlp_ratio: int = 4
    ln_eps: float = 1e-6
    patch: int = PATCH
    pos_grid: int = POS_GRID

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def mlp(self) -> int:
        return self.hidden * self.mlp_ratio


MODELS = {
    # KAT-tiny: known-answer-test size, same structure (SURVEY.md section 8c fixtures (i))
    "tiny": ModelConfig("tiny", 128, 2, 4, (1, 2, 3, 4), (16, 32, 64, 128), 32),
    "vits": ModelConfig("vits", 384, 6, 12, (3, 6, 9, 12), (48, 96, 192, 384), 64),
    "vitb": ModelConfig("vitb", 768, 12, 12, (3, 6, 9, 12), (96, 192, 384, 768), 128),
    "vitl": ModelConfig("vitl", 1024, 16, 24, (5, 12, 18, 24), (256, 512, 1024, 1024), 256),
}

# reference utils.py:734-736 model ids -> architecture
MODEL_IDS = {
    "depth-anything/Depth-Anything-V2-Small-hf": "vits",
    "depth-anything/Depth-Anything-V2-Base-hf": "vitb",
    "depth-anything/Depth-Anything-V2-Large-hf": "vitl",
}
# metric variants (reference utils.py:761-769): same architecture, head ends in sigmoid * max_depth
# (HF config.json: depth_estimation_type "metric", max_depth 20 indoor / 80 outdoor)
METRIC_MODEL_IDS = {
    f"depth-anything/Depth-Anything-V2-Metric-{scene}-{size}-hf": (arch, max_depth)
    for scene, max_depth in (("Indoor", 20.0), ("Outdoor", 80.0))
    for size, arch in (("Small", "vits"), ("Base", "vitb"), ("Large", "vitl"))
}


def is_metric_id(model_id: str) -> bool:
    """reference depth.py:666: inversion in normalize() is keyed off the model id."""
    return any(k in model_id.lower() for k in ("metric", "kitti", "nyu", "depth-ai", "da3"))


@dataclass
class PipelineParams:
    """Constants the hot path reads (reference utils.py:837-859, 900)."""
    depth_resolution: int = 518        # settings "Depth Resolution" (default 336; BASELINE uses 518)
    foreground_scale: float = 0.05     # yaml 0.5 / 10            utils.py:858
    aa_strength: float = 4.0           # yaml 2 * 2               utils.py:859
    gamma: float = 1.45                # depth.py:775
    percentile: float = 2.0            # depth.py:816
    subsample_cap: int = 6144          # depth.py:816
    ema_alpha: float = 0.9             # depth.py:1889
    ipd: float = 0.064                 # utils.py:851
    depth_strength: float = 4.0        # settings "Depth Strength" utils.py:850
    convergence: float = 0.0           # utils.py:852
    display_mode: str = "Half-SBS"     # utils.py:840
    fill_16_9: bool = True             # utils.py:900
    metric: bool = False               # is_metric(), depth.py:666-669 (1/d inversion in normalize)
    mean: Tuple[float, float, float] = IMAGENET_MEAN
    std: Tuple[float, float, float] = IMAGENET_STD
    # which branch of _resize_patch_aligned_t (reference depth.py:676-706) the pre-process follows:
    #   "bilinear"   -- the CPU / DirectML branch (::stride decimation + bilinear): what the reference's CPU path computes,
    #                   the path BASELINE.json's parity bar names (default);
    #   "bicubic_aa" -- the IS_CUDA branch (depth.py:698-699; true on a ROCm device): bicubic + antialias from the full frame
    resample: str = "bilinear"
    # reference CAPTURE_MODE (utils.py; settings "Capture Mode"): "Window" makes get_patch_size() return None (depth.py:531-538), so
    # predict_depth takes the fixed-square branch -- plain bilinear of the full frame to depth_resolution x depth_resolution
    # (depth.py:1937-1946) -- instead of the aspect-preserving patch-aligned resize ("Monitor")
    capture_mode: str = "Monitor"

    @property
    def square_input(self) -> bool:
        return self.capture_mode == "Window"

    def asdict(self):
        return asdict(self)


def nearest_multiple(x: int, p: int) -> int:
    """Ties go UP (reference depth.py:683-686)."""
    down = (x // p) * p
    up = down + p
    return up if abs(up - x) <= abs(x - down) else down


def engine_shape(h: int, w: int, target: int, patch: int = PATCH, square: bool = False):
    """Model-input size and the CPU-branch decimation stride for an h x w frame.

    square=True: the fixed-square branch of predict_depth (get_patch_size() is None, reference
    depth.py:1937-1946): target x target from the full frame, no decimation.

    Restates the integer logic of ``_resize_patch_aligned_t`` (reference
    depth.py:676-706): longest side -> target, each dim to the nearest patch
    multiple (ties up); CPU branch pre-decimates by ``longest // (2*target)``.
    Returns (new_h, new_w, stride) with stride >= 1.
    """
    if square:
        if target % patch:
            raise ValueError(f"fixed-square input: Depth Resolution {target} is not a multiple of the patch size {patch}")
        return target, target, 1
    longest = max(h, w)
    scale = target / float(longest) if longest != target else 1.0
    sh = max(1, int(round(h * scale)))
    sw = max(1, int(round(w * scale)))
    new_h = max(1, nearest_multiple(sh, patch))
    new_w = max(1, nearest_multiple(sw, patch))
    stride = longest // (target * 2)
    if stride < 1:
        stride = 1
    return new_h, new_w, stride
