"""Seeded synthetic frames (no datasets offline).

S1 ``noise_frame``      : uniform uint8 noise -- worst case for the warp's gathers (throughput runs).
S2 ``structured_frame`` : smooth gradients + filled rectangles / discs, so the depth field and the
                          percentile normalisation downstream are well conditioned (parity runs).
``smooth_depth``        : a smooth [0,1] depth map with a few step edges, for warp-only tests.
(SURVEY.md section 8d "Synthetic inputs".)
"""
from __future__ import annotations

import numpy as np


def noise_frame(h: int, w: int, seed: int = 0) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def structured_frame(h: int, w: int, seed: int = 0) -> np.ndarray:
    g = np.random.default_rng(1000 + seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    u, v = xx / max(w - 1, 1), yy / max(h - 1, 1)
    img = np.stack([60 + 120 * u, 40 + 150 * v, 200 - 100 * u * v], -1).astype(np.float32)
    img += 12.0 * np.sin(2 * np.pi * (3 * u + 2 * v))[..., None]
    for k in range(5):
        col = g.integers(0, 256, 3).astype(np.float32)
        cx, cy = g.uniform(0.15, 0.85) * w, g.uniform(0.15, 0.85) * h
        rx, ry = g.uniform(0.05, 0.2) * w, g.uniform(0.05, 0.2) * h
        if k % 2 == 0:
            m = (np.abs(xx - cx) < rx) & (np.abs(yy - cy) < ry)
        else:
            m = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 < 1.0
        img[m] = col
    img += g.normal(0.0, 2.0, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def smooth_depth(h: int, w: int, seed: int = 0) -> np.ndarray:
    g = np.random.default_rng(2000 + seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    u, v = xx / max(w - 1, 1), yy / max(h - 1, 1)
    d = 0.35 + 0.3 * v + 0.15 * np.sin(2 * np.pi * (1.5 * u + 0.5 * v))
    for _ in range(3):
        cx, cy = g.uniform(0.2, 0.8), g.uniform(0.2, 0.8)
        r = g.uniform(0.08, 0.2)
        d = np.where((u - cx) ** 2 + (v - cy) ** 2 < r * r, g.uniform(0.6, 1.0), d)
    return np.clip(d, 0.0, 1.0).astype(np.float32)


def dibr_scene(h: int, w: int, seed: int, kind: str = "boxes"):
    """(rgb uint8 [h,w,3], depth float32 [h,w] in 0..1) for the viewer-shader warp (f1): structured_frame + a smooth depth map and,
    for kind "boxes", three near rectangles with hard edges -- disocclusions on both sides, the in-painting's work; "smooth" stays
    under the shader's 0.04 discontinuity threshold.  The recipe behind tests/golden/dibr.npz (make_golden_dibr.py)."""
    img = structured_frame(h, w, seed)
    dep = (0.35 + 0.6 * smooth_depth(h, w, seed)).astype(np.float32)
    if kind == "boxes":
        rng = np.random.default_rng(seed)
        for k in range(3):
            y0, x0 = int(rng.integers(0, h * 2 // 3)), int(rng.integers(0, w * 2 // 3))
            hh, ww = int(rng.integers(h // 8, h // 3)), int(rng.integers(w // 10, w // 3))
            dep[y0:y0 + hh, x0:x0 + ww] = np.float32(0.05 + 0.1 * k)
    return img, np.clip(dep, 0, 1).astype(np.float32)
