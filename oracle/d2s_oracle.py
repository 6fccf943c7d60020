"""CPU oracle for the desktop2stereo hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A numpy (float32) restatement of the reference's per-frame path
    predict_depth  (reference depth.py:1897-2025)
    make_sbs / make_sbs_core  (reference depth.py:2122-2231)
and of the Depth-Anything-v2 model arithmetic the reference obtains from the third-party
dependency ``transformers`` (pinned 4.56.2 in reference requirements.txt:5; call sites
depth.py:14, 1649-1662, 1778).  The model code is NOT under /root/reference; it is restated here
from the published architecture (DINOv2 ViT + DPT neck/head: HF modeling_dinov2.py /
modeling_depth_anything.py) and pinned by golden vectors captured from the reference run in the
build container (tests/golden/make_golden.py -> tests/golden/*.npz).  Parity status: PINNED by
those generated fixtures; the reference itself ships no tests or golden vectors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (desktop2stereo_amd) never does.

Every function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np

try:  # exact erf for GELU
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float32])

F32 = np.float32


# ----------------------------------------------------------------------------------------------
# integer shape logic
# ----------------------------------------------------------------------------------------------
def nearest_multiple(x: int, p: int) -> int:
    """reference depth.py:683-686 (ties go up)."""
    down = (x // p) * p
    up = down + p
    return up if abs(up - x) <= abs(x - down) else down


def engine_shape(h: int, w: int, target: int, patch: int = 14) -> Tuple[int, int, int]:
    """reference depth.py:677-689, 703: (new_h, new_w, cpu-branch decimation stride)."""
    longest = max(h, w)
    scale = target / float(longest) if longest != target else 1.0
    sh = max(1, int(round(h * scale)))
    sw = max(1, int(round(w * scale)))
    new_h = max(1, nearest_multiple(sh, patch))
    new_w = max(1, nearest_multiple(sw, patch))
    stride = max(1, longest // (target * 2))
    return new_h, new_w, stride


# ----------------------------------------------------------------------------------------------
# resampling primitives (torch.nn.functional.interpolate / grid_sample semantics)
# ----------------------------------------------------------------------------------------------
def _linear_taps(in_size: int, out_size: int, align_corners: bool):
    """Source taps of torch's (non-antialiased) linear interpolation along one axis.

    align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0, scale = in/out (float32)
    align_corners=True : src = dst*(in-1)/(out-1)
    i1 = min(i0+1, in-1); weights (1-l, l).  (ATen UpSample.h area_pixel_compute_source_index)
    """
    dst = np.arange(out_size, dtype=F32)
    if align_corners:
        scale = F32(in_size - 1) / F32(out_size - 1) if out_size > 1 else F32(0)
        src = dst * scale
    else:
        scale = F32(in_size) / F32(out_size)
        src = scale * (dst + F32(0.5)) - F32(0.5)
        src = np.maximum(src, F32(0))
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, in_size - 1)
    i1 = np.minimum(i0 + 1, in_size - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    l0 = (F32(1) - l1).astype(F32)
    return i0, i1, l0, l1


def bilinear_resize(x: np.ndarray, out_h: int, out_w: int, align_corners: bool) -> np.ndarray:
    """F.interpolate(mode='bilinear') on [..., H, W] float32."""
    x = np.asarray(x, dtype=F32)
    H, W = x.shape[-2:]
    y0, y1, wy0, wy1 = _linear_taps(H, out_h, align_corners)
    x0, x1, wx0, wx1 = _linear_taps(W, out_w, align_corners)
    top = x[..., y0, :]
    bot = x[..., y1, :]
    wy0 = wy0[:, None]
    wy1 = wy1[:, None]
    # ATen: w_y0*(w_x0*a + w_x1*b) + w_y1*(w_x0*c + w_x1*d)
    t = top[..., x0] * wx0 + top[..., x1] * wx1
    b = bot[..., x0] * wx0 + bot[..., x1] * wx1
    return (wy0 * t + wy1 * b).astype(F32)


def _cubic_coeffs(t: np.ndarray, A: float = -0.75):
    """ATen get_cubic_upsample_coefficients (cubic convolution, A=-0.75)."""
    t = t.astype(F32)
    A = F32(A)

    def c1(x):  # |x| <= 1
        return ((A + F32(2)) * x - (A + F32(3))) * x * x + F32(1)

    def c2(x):  # 1 < |x| < 2
        return ((A * x - F32(5) * A) * x + F32(8) * A) * x - F32(4) * A
    return c2(t + F32(1)), c1(t), c1(F32(1) - t), c2(F32(2) - t)


def bicubic_resize(x: np.ndarray, out_h: int, out_w: int, scale_factor=None) -> np.ndarray:
    """F.interpolate(mode='bicubic', align_corners=False) on [C,H,W] float32 (no antialias).
    scale_factor=(sy, sx): the caller-supplied factors enter the coordinate map as float32(1/s)
    (ATen area_pixel_compute_scale with `scales`), as in the vendored DINOv2 of Video-Depth-Anything
    (reference models/video_depth_anything/dinov2.py:179-210)."""
    x = np.asarray(x, dtype=F32)
    C, H, W = x.shape
    fh = F32(1.0 / scale_factor[0]) if scale_factor is not None else None
    fw = F32(1.0 / scale_factor[1]) if scale_factor is not None else None

    def taps(n_in, n_out, forced=None):
        scale = forced if forced is not None else F32(n_in) / F32(n_out)
        src = scale * (np.arange(n_out, dtype=F32) + F32(0.5)) - F32(0.5)
        i = np.floor(src)
        t = (src - i).astype(F32)
        i = i.astype(np.int64)
        idx = np.stack([np.clip(i + k, 0, n_in - 1) for k in (-1, 0, 1, 2)], 0)
        return idx, np.stack(_cubic_coeffs(t), 0).astype(F32)
    iy, cy = taps(H, out_h, fh)
    ix, cx = taps(W, out_w, fw)
    # ATen order: for each of 4 rows interpolate along x, then along y
    rows = []
    for k in range(4):
        r = x[:, iy[k], :]                                         # [C,out_h,W]
        acc = (r[:, :, ix[0]] * cx[0] + r[:, :, ix[1]] * cx[1]
               + r[:, :, ix[2]] * cx[2] + r[:, :, ix[3]] * cx[3])
        rows.append(acc.astype(F32))
    out = (rows[0] * cy[0][None, :, None] + rows[1] * cy[1][None, :, None]
           + rows[2] * cy[2][None, :, None] + rows[3] * cy[3][None, :, None])
    return out.astype(F32)


# ----------------------------------------------------------------------------------------------
# A2-A4  ingest / resize / normalise
# ----------------------------------------------------------------------------------------------
def resize_patch_aligned(img_chw: np.ndarray, target: int, patch: int = 14, cuda_branch: bool = False) -> np.ndarray:
    """reference depth.py:676-706.  CPU / DirectML branch (default, :700-706): strided decimation then bilinear
    (align_corners=False, no antialias) in float32.  cuda_branch=True (:698-699, what the reference runs when
    IS_CUDA -- which includes ROCm devices): ONE F.interpolate(bicubic, align_corners=False, antialias=True) from the
    full frame, no decimation, no clamp (the cubic lobes overshoot 0..255).  img_chw: [3,H,W] uint8 or float."""
    _, h, w = img_chw.shape
    new_h, new_w, stride = engine_shape(h, w, target, patch)
    x = np.asarray(img_chw)
    if new_h == h and new_w == w:
        return x.astype(F32)
    if cuda_branch:
        return separable_aa_resize(x.astype(F32), new_h, new_w, cubic=True)
    if stride > 1:
        x = x[:, ::stride, ::stride]
    return bilinear_resize(x.astype(F32), new_h, new_w, align_corners=False)


def resize_fixed_square(img_chw: np.ndarray, target: int) -> np.ndarray:
    """reference depth.py:1937-1946: the branch predict_depth takes when get_patch_size() is None (CAPTURE_MODE == "Window",
    depth.py:531-538) -- F.interpolate(bilinear, align_corners=False, no antialias) of the full frame to target x target, on
    every device; a frame that already is target x target passes through."""
    _, h, w = img_chw.shape
    x = np.asarray(img_chw).astype(F32)
    if (h, w) == (target, target):
        return x
    return bilinear_resize(x, target, target, align_corners=False)


def normalise(x: np.ndarray, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> np.ndarray:
    """reference depth.py:1931, 1946-1948: x/255 then (x-mean)/std."""
    m = np.asarray(mean, dtype=F32).reshape(3, 1, 1)
    s = np.asarray(std, dtype=F32).reshape(3, 1, 1)
    return ((x.astype(F32) / F32(255.0)) - m) / s


# ----------------------------------------------------------------------------------------------
# A5-A9  Depth-Anything-v2 model (HF DepthAnythingForDepthEstimation arithmetic)
# ----------------------------------------------------------------------------------------------
def layer_norm(x, g, b, eps=1e-6):
    mu = x.mean(-1, keepdims=True, dtype=F32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=F32)
    return ((x - mu) / np.sqrt(var + F32(eps)) * g + b).astype(F32)


def gelu_erf(x):
    return (F32(0.5) * x * (F32(1.0) + _erf(x * F32(0.7071067811865476)).astype(F32))).astype(F32)


def interpolate_pos_embed(pos: np.ndarray, gh: int, gw: int, grid: int = 37, offset: float = 0.0) -> np.ndarray:
    """HF Dinov2Embeddings.interpolate_pos_encoding: bicubic (align_corners=False) resample of
    the [grid,grid,D] patch position table to [gh,gw,D]; cls row kept.  pos: [1, 1+grid*grid, D].
    offset=0.1: the vendored DINOv2 variant (scale_factor=((gh+0.1)/grid, (gw+0.1)/grid))."""
    D = pos.shape[-1]
    if gh == grid and gw == grid:
        return pos[0].astype(F32)
    cls = pos[0, :1]
    tab = pos[0, 1:].reshape(grid, grid, D).transpose(2, 0, 1)     # [D,grid,grid]
    sf = ((gh + offset) / grid, (gw + offset) / grid) if offset else None
    tab = bicubic_resize(tab, gh, gw, sf)                            # [D,gh,gw]
    tab = tab.transpose(1, 2, 0).reshape(gh * gw, D)
    return np.concatenate([cls, tab], 0).astype(F32)


def conv2d(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride=1, pad=0) -> np.ndarray:
    """nn.Conv2d on [C,H,W] float32 via im2col + matmul (cross-correlation)."""
    x = np.asarray(x, dtype=F32)
    Co, Ci, kh, kw = w.shape
    C, H, W = x.shape
    if pad:
        x = np.pad(x, ((0, 0), (pad, pad), (pad, pad)))
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if kh == 1 and kw == 1 and stride == 1:
        out = w.reshape(Co, Ci) @ x.reshape(Ci, -1)
    else:
        cols = np.empty((Ci, kh, kw, Ho, Wo), dtype=F32)
        for i in range(kh):
            for j in range(kw):
                cols[:, i, j] = x[:, i:i + stride * Ho:stride, j:j + stride * Wo:stride]
        out = w.reshape(Co, -1) @ cols.reshape(Ci * kh * kw, Ho * Wo)
    out = out.reshape(Co, Ho, Wo)
    if b is not None:
        out = out + b.reshape(-1, 1, 1)
    return out.astype(F32)


def conv_transpose_k_eq_s(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """nn.ConvTranspose2d(kernel=stride=k, padding=0): every output pixel has exactly one tap.
    x [Ci,H,W], w [Ci,Co,k,k] -> [Co,H*k,W*k]."""
    Ci, Co, k, _ = w.shape
    _, H, W = x.shape
    y = np.einsum("ihw,iokl->ohkwl", x.astype(F32), w.astype(F32), optimize=True)
    y = y.reshape(Co, H * k, W * k) + b.reshape(-1, 1, 1)
    return y.astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


class DepthAnythingOracle:
    """float32 numpy forward of HF DepthAnythingForDepthEstimation.
    max_depth == 0: relative head (conv3 -> ReLU); max_depth > 0: metric head, sigmoid(conv3) * max_depth
    (HF DepthAnythingDepthEstimationHead with depth_estimation_type == "metric")."""

    def __init__(self, cfg, weights: Dict[str, np.ndarray], max_depth: float = 0.0):
        self.cfg = cfg
        self.max_depth = float(max_depth)
        self.w = {k: np.asarray(v, dtype=F32) for k, v in weights.items()}

    # -- backbone ------------------------------------------------------------------------------
    def embeddings(self, x: np.ndarray) -> np.ndarray:
        """HF Dinov2Embeddings.forward: conv k=s=14 patch embed, prepend cls, add pos-embed."""
        cfg, w = self.cfg, self.w
        p = cfg.patch
        _, H, W = x.shape
        gh, gw = H // p, W // p
        patches = x.reshape(3, gh, p, gw, p).transpose(1, 3, 0, 2, 4).reshape(gh * gw, 3 * p * p)
        pw = w["backbone.embeddings.patch_embeddings.projection.weight"].reshape(cfg.hidden, -1)
        tok = patches @ pw.T + w["backbone.embeddings.patch_embeddings.projection.bias"]
        tok = np.concatenate([w["backbone.embeddings.cls_token"][0], tok], 0)
        pos = interpolate_pos_embed(w["backbone.embeddings.position_embeddings"], gh, gw, cfg.pos_grid,
                                    getattr(self, "pos_offset", 0.0))
        return (tok + pos).astype(F32)

    def layer(self, i: int, x: np.ndarray) -> np.ndarray:
        """HF Dinov2Layer.forward (LN eps 1e-6, exact GELU, LayerScale)."""
        cfg, w = self.cfg, self.w
        p = f"backbone.encoder.layer.{i}."
        N, D = x.shape
        h = layer_norm(x, w[p + "norm1.weight"], w[p + "norm1.bias"], cfg.ln_eps)
        q = h @ w[p + "attention.attention.query.weight"].T + w[p + "attention.attention.query.bias"]
        k = h @ w[p + "attention.attention.key.weight"].T + w[p + "attention.attention.key.bias"]
        v = h @ w[p + "attention.attention.value.weight"].T + w[p + "attention.attention.value.bias"]
        nh, hd = cfg.heads, cfg.head_dim
        q = q.reshape(N, nh, hd).transpose(1, 0, 2)
        k = k.reshape(N, nh, hd).transpose(1, 0, 2)
        v = v.reshape(N, nh, hd).transpose(1, 0, 2)
        s = (q @ k.transpose(0, 2, 1)) * F32(hd ** -0.5)
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        a = e / e.sum(-1, keepdims=True, dtype=F32)
        o = (a @ v).transpose(1, 0, 2).reshape(N, D)
        o = o @ w[p + "attention.output.dense.weight"].T + w[p + "attention.output.dense.bias"]
        x = x + o * w[p + "layer_scale1.lambda1"]
        h = layer_norm(x, w[p + "norm2.weight"], w[p + "norm2.bias"], cfg.ln_eps)
        h = gelu_erf(h @ w[p + "mlp.fc1.weight"].T + w[p + "mlp.fc1.bias"])
        h = h @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"]
        return (x + h * w[p + "layer_scale2.lambda1"]).astype(F32)

    def backbone(self, x: np.ndarray, taps: Optional[dict] = None):
        """HF Dinov2Backbone.forward: hidden states after out_indices layers, shared final LN."""
        cfg, w = self.cfg, self.w
        h = self.embeddings(x)
        if taps is not None:
            taps["embeddings"] = h
        feats = []
        for i in range(cfg.layers):
            h = self.layer(i, h)
            if taps is not None:
                taps[f"layer{i + 1}"] = h
            if (i + 1) in cfg.out_indices:
                feats.append(layer_norm(h, w["backbone.layernorm.weight"], w["backbone.layernorm.bias"], cfg.ln_eps))
        return feats

    # -- neck + head ---------------------------------------------------------------------------
    def _rcu(self, p: str, x: np.ndarray) -> np.ndarray:
        """HF DepthAnythingPreActResidualLayer."""
        w = self.w
        h = conv2d(relu(x), w[p + "convolution1.weight"], w[p + "convolution1.bias"], 1, 1)
        h = conv2d(relu(h), w[p + "convolution2.weight"], w[p + "convolution2.bias"], 1, 1)
        return h + x

    def neck_head(self, feats, gh: int, gw: int, taps: Optional[dict] = None, hooks: Optional[dict] = None) -> np.ndarray:
        cfg, w = self.cfg, self.w
        maps = []
        for i, f in enumerate(feats):
            p = f"neck.reassemble_stage.layers.{i}."
            x = f[1:].reshape(gh, gw, cfg.hidden).transpose(2, 0, 1)
            x = conv2d(x, w[p + "projection.weight"], w[p + "projection.bias"])
            if i == 0 or i == 1:
                x = conv_transpose_k_eq_s(x, w[p + "resize.weight"], w[p + "resize.bias"])
            elif i == 3:
                x = conv2d(x, w[p + "resize.weight"], w[p + "resize.bias"], stride=2, pad=1)
            if hooks and f"layer_{i + 1}" in hooks:                # VDA: temporal module on layer_3 / layer_4
                x = hooks[f"layer_{i + 1}"](x)
            x = conv2d(x, w[f"neck.convs.{i}.weight"], None, 1, 1)
            if taps is not None:
                taps[f"neck_feat{i}"] = x
            maps.append(x)
        maps = maps[::-1]                                           # HF FeatureFusionStage: deep -> shallow
        fused = None
        for idx, m in enumerate(maps):
            p = f"neck.fusion_stage.layers.{idx}."
            if fused is None:
                h = m
            else:
                h = fused + self._rcu(p + "residual_layer1.", m)
            h = self._rcu(p + "residual_layer2.", h)
            if idx != len(maps) - 1:
                oh, ow = maps[idx + 1].shape[1:]
            else:
                oh, ow = h.shape[1] * 2, h.shape[2] * 2
            h = bilinear_resize(h, oh, ow, align_corners=True)
            fused = conv2d(h, w[p + "projection.weight"], w[p + "projection.bias"])
            if hooks and f"path_{4 - idx}" in hooks:               # VDA: temporal module on path_4 / path_3
                fused = hooks[f"path_{4 - idx}"](fused)
            if taps is not None:
                taps[f"fused{idx}"] = fused
        h = conv2d(fused, w["head.conv1.weight"], w["head.conv1.bias"], 1, 1)
        h = bilinear_resize(h, gh * cfg.patch, gw * cfg.patch, align_corners=True)
        h = relu(conv2d(h, w["head.conv2.weight"], w["head.conv2.bias"], 1, 1))
        h = conv2d(h, w["head.conv3.weight"], w["head.conv3.bias"])
        if getattr(self, "max_depth", 0.0) > 0.0:
            h = (F32(1) / (F32(1) + np.exp(-h.astype(F32)))).astype(F32) * F32(self.max_depth)
        else:
            h = relu(h)
        return h[0].astype(F32)

    def forward(self, x: np.ndarray, taps: Optional[dict] = None, hooks: Optional[dict] = None) -> np.ndarray:
        """x: normalised [3,h,w] float32 -> predicted_depth [h,w]."""
        p = self.cfg.patch
        gh, gw = x.shape[1] // p, x.shape[2] // p
        feats = self.backbone(x, taps)
        return self.neck_head(feats, gh, gw, taps, hooks)


# ----------------------------------------------------------------------------------------------
# A10-A13  post-process
# ----------------------------------------------------------------------------------------------
def percentile_bounds(d: np.ndarray, percentile=2.0, subsample_cap=6144):
    """reference depth.py:850-863 + 784-794: subsample every ceil(n/cap)-th value, then the
    tail-th smallest / largest (no lerp)."""
    v = d.reshape(-1)
    if v.size <= 10:
        return F32(0), F32(0)
    if v.size > subsample_cap:
        step = (v.size + subsample_cap - 1) // subsample_cap
        v = v[::step]
    n = v.size
    lo_q = max(0.0, min(1.0, float(percentile) / 100.0))
    tail = min(n, max(1, int(round(lo_q * (n - 1))) + 1))
    if tail == n:
        return v.min(), v.max()
    s = np.sort(v)
    return s[tail - 1], s[n - tail]


def normalize_depth(d: np.ndarray, percentile=2.0, subsample_cap=6144, metric=False) -> np.ndarray:
    """reference depth.py:816-867.  metric (is_metric(), :844-847): inv = 1/max(d,1e-12) on d > 0 (others keep d),
    order statistics over the valid values only (row-major compaction), <= 10 valid values -> bounds (0, 0)."""
    d = np.asarray(d, dtype=F32)
    if metric:
        valid = d > 0
        with np.errstate(divide="ignore"):
            d = np.where(valid, F32(1) / np.maximum(d, F32(1e-12)), d).astype(F32)
        dmin, dmax = percentile_bounds(d[valid], percentile, subsample_cap)
    else:
        dmin, dmax = percentile_bounds(d, percentile, subsample_cap)
    denom = np.maximum(F32(dmax) - F32(dmin), F32(1e-6))
    return np.clip((d - F32(dmin)) / denom, F32(0), F32(1)).astype(F32)


def apply_gamma(d, gamma=1.45):
    """reference depth.py:775-776."""
    return np.power(d.astype(F32), F32(gamma)).astype(F32)


def apply_foreground_scale(d, scale, mid=0.5, eps=1e-6):
    """reference depth.py:709-736."""
    d = np.clip(d.astype(F32), F32(0), F32(1))
    if abs(scale) < eps:
        return d
    exponent = F32(1.0 / (1.0 + scale))
    dist = d - F32(mid)
    out = F32(mid) + np.sign(dist) * np.power(np.abs(dist), exponent)
    return np.clip(out, F32(0), F32(1)).astype(F32)


def gaussian_taps(strength: float) -> np.ndarray:
    """reference depth.py:746-758: k = int(3s)|1, sigma = 0.5 s, normalised float32 taps."""
    k = int(3 * strength) | 1
    sigma = 0.5 * strength
    c = np.arange(k, dtype=F32) - F32(k // 2)
    g = np.exp(-(c ** 2) / F32(2 * sigma ** 2)).astype(F32)
    return (g / g.sum(dtype=F32)).astype(F32)


def anti_alias(d: np.ndarray, strength: float) -> np.ndarray:
    """reference depth.py:740-765: separable Gaussian, zero padding, horizontal then vertical."""
    k = int(3 * strength) | 1
    if k < 3:
        return d
    g = gaussian_taps(strength)
    r = k // 2
    H, W = d.shape
    x = np.pad(d.astype(F32), ((0, 0), (r, r)))
    h = np.zeros((H, W), F32)
    for i in range(k):
        h += g[i] * x[:, i:i + W]
    x = np.pad(h, ((r, r), (0, 0)))
    v = np.zeros((H, W), F32)
    for i in range(k):
        v += g[i] * x[i:i + H, :]
    return v


def post_process_depth(d, foreground_scale=0.05, aa_strength=4.0, gamma=1.45, metric=False):
    """reference depth.py:806-814."""
    d = normalize_depth(d, metric=metric)
    d = apply_gamma(d, gamma)
    d = apply_foreground_scale(d, foreground_scale)
    return anti_alias(d, aa_strength)


class DepthStabilizer:
    """reference depth.py:1865-1887: prev = d on first frame / shape change, else
    prev <- lerp(prev, d, 1-alpha)."""

    def __init__(self, alpha=0.9):
        self.alpha = alpha
        self.prev = None

    def __call__(self, d):
        if self.prev is None or self.prev.shape != d.shape:
            self.prev = d.astype(F32).copy()
            return d
        wgt = F32(1.0 - self.alpha)
        # torch lerp (weight < 0.5): start + weight*(end-start)
        self.prev = (self.prev + wgt * (d.astype(F32) - self.prev)).astype(F32)
        return self.prev


def upsample_depth(d: np.ndarray, h: int, w: int) -> np.ndarray:
    """reference depth.py:1999-2004: bilinear, align_corners=False."""
    return bilinear_resize(d, h, w, align_corners=False)


# ----------------------------------------------------------------------------------------------
# A14  stereo warp
# ----------------------------------------------------------------------------------------------
def _torch_linspace(n: int) -> np.ndarray:
    """torch.linspace(-1, 1, n) float32 (ATen: symmetric evaluation from both ends)."""
    start, end = F32(-1.0), F32(1.0)
    step = (end - start) / F32(n - 1)
    i = np.arange(n)
    lo = start + step * i.astype(F32)
    hi = end - step * (n - 1 - i).astype(F32)
    return np.where(i < n // 2, lo, hi).astype(F32)


def _reflect(x: np.ndarray, span: F32) -> np.ndarray:
    """ATen grid_sampler reflect_coordinates(in, twice_low=0, twice_high=2*span)."""
    x = np.abs(x)
    extra = np.fmod(x, span).astype(F32)
    flips = np.floor(x / span)
    return np.where(np.fmod(flips, 2) == 0, extra, span - extra).astype(F32)


def grid_sample_rows(img: np.ndarray, shift_px_sign: float, shifts: np.ndarray) -> np.ndarray:
    """F.grid_sample(bilinear, reflection, align_corners=True) for the reference's grid
    (xs + sign*shift_norm, ys) -- reference depth.py:2152-2160 -- following the float32
    normalised-coordinate round trip exactly.  img [C,H,W] float32, shifts [H,W] (pixels)."""
    C, H, W = img.shape
    xs = _torch_linspace(W)[None, :]
    ys = _torch_linspace(H)[:, None]
    shift_norm = (shifts.astype(F32) * F32(2.0 / (W - 1))).astype(F32)
    gx = (xs + F32(shift_px_sign) * shift_norm).astype(F32)
    gy = np.broadcast_to(ys, (H, W)).astype(F32)
    ix = ((gx + F32(1)) / F32(2)) * F32(W - 1)
    iy = ((gy + F32(1)) / F32(2)) * F32(H - 1)
    ix = np.clip(_reflect(ix, F32(W - 1)), F32(0), F32(W - 1))
    iy = np.clip(_reflect(iy, F32(H - 1)), F32(0), F32(H - 1))
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    wx1 = (ix - x0).astype(F32)
    wy1 = (iy - y0).astype(F32)
    wx0 = F32(1) - wx1
    wy0 = F32(1) - wy1
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    x1 = x0 + 1
    y1 = y0 + 1

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok[None], v, F32(0))
    out = (tap(y0, x0) * (wx0 * wy0)[None] + tap(y0, x1) * (wx1 * wy0)[None]
           + tap(y1, x0) * (wx0 * wy1)[None] + tap(y1, x1) * (wx1 * wy1)[None])
    return out.astype(F32)


def pad_to_aspect(t: np.ndarray, ratio=(16, 9)) -> np.ndarray:
    """reference depth.py:2106-2119."""
    _, h, w = t.shape
    r_img, r_t = w / h, ratio[0] / ratio[1]
    if abs(r_img - r_t) < 1e-3:
        return t
    if r_img > r_t:
        new_h = int(round(w / r_t))
        top = (new_h - h) // 2
        return np.pad(t, ((0, 0), (top, new_h - h - top), (0, 0)))
    new_w = int(round(h * r_t))
    left = (new_w - w) // 2
    return np.pad(t, ((0, 0), (0, 0), (left, new_w - w - left)))


def make_sbs_core(rgb: np.ndarray, depth: np.ndarray, ipd_uv=0.064, depth_ratio=2.0,
                  display_mode="Half-SBS", fill_16_9=False, convergence=0.0) -> np.ndarray:
    """reference depth.py:2122-2184 (grid_sample path).  rgb [C,H,W] float 0..255, depth [H,W].
    Returns [C,H',W'] float32 0..255."""
    rgb = np.asarray(rgb, dtype=F32)
    C, H, W = rgb.shape
    img = np.clip(rgb, F32(0), F32(255))
    d = np.asarray(depth, dtype=F32) - F32(convergence)
    inv = -d * F32(depth_ratio)
    max_px = ipd_uv * W
    shifts = (inv * F32(max_px)).astype(F32) * F32(0.05)
    left = grid_sample_rows(img, +1.0, shifts)
    right = grid_sample_rows(img, -1.0, shifts)
    if fill_16_9:
        left = pad_to_aspect(left)
        right = pad_to_aspect(right)
    if display_mode in ("Half-TAB", "Full-TAB"):
        out = np.concatenate([left, right], 1)
    else:
        out = np.concatenate([left, right], 2)
    if display_mode not in ("Full-SBS", "Full-TAB"):
        hh, ww = left.shape[1:]
        if display_mode == "Half-SBS":                              # F.interpolate(mode='area'): 2:1 mean
            out = (out[:, :, 0::2] + out[:, :, 1::2]) * F32(0.5)
        else:
            out = (out[:, 0::2, :] + out[:, 1::2, :]) * F32(0.5)
        assert out.shape[1:] == (hh, ww)
    return np.clip(out, F32(0), F32(255)).astype(F32)


def to_u8(x: np.ndarray) -> np.ndarray:
    """round-half-even + saturate: the implicit convertTo(CV_8U) of the reference's sink
    (cv2.imencode of the float32 frame, reference streamer.py:250-252)."""
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------------------------
# A1 process() / A15 overlay_fps(): the rows either side of the path
# ----------------------------------------------------------------------------------------------
def _aa_weights(in_size: int, out_size: int, cubic: bool = False):
    """ATen _upsample_bilinear2d_aa / _upsample_bicubic2d_aa per-output taps (third-party torch; UpSampleKernel.cpp
    _compute_indices_min_size_weights_aa): scale = in/out, support = (interp_size/2)*scale when down-scaling
    (interp_size 2: triangle filter; 4: Keys cubic with a = -0.5, HelperInterpCubic::aa_filter), center = scale*(i+0.5),
    taps [xmin, xmin+xsize), weights normalised to sum 1.  float32 like ATen's opmath for float tensors."""
    scale = F32(in_size) / F32(out_size)
    half = F32(2.0 if cubic else 1.0)
    support = F32(half * scale) if scale >= 1 else half
    invscale = F32(1) / scale if scale >= 1 else F32(1)
    taps = []
    for i in range(out_size):
        center = F32(scale * F32(i + 0.5))
        xmin = max(int(float(F32(center - support)) + 0.5), 0)
        xsize = min(int(float(F32(center + support)) + 0.5), in_size) - xmin
        x = np.abs(((np.arange(xsize, dtype=F32) + F32(xmin)) - center + F32(0.5)) * invscale).astype(F32)
        if cubic:
            a = F32(-0.5)
            w1 = ((a + F32(2)) * x - (a + F32(3))) * x * x + F32(1)                      # |x| < 1
            w2 = ((a * x - F32(5) * a) * x + F32(8) * a) * x - F32(4) * a                # 1 <= |x| < 2
            w = np.where(x < 1, w1, np.where(x < 2, w2, F32(0))).astype(F32)
        else:
            w = np.where(x < 1, F32(1) - x, F32(0)).astype(F32)
        taps.append((xmin, (w / w.sum(dtype=F32)).astype(F32)))
    return taps


def separable_aa_resize(x: np.ndarray, nh: int, nw: int, cubic: bool) -> np.ndarray:
    """F.interpolate(mode=bilinear|bicubic, align_corners=False, antialias=True) on [C,H,W] float32: ATen's separable
    kernel, horizontal pass then vertical pass (separable_upsample_generic_Nd_kernel_impl), taps accumulated in order."""
    C, H0, W0 = x.shape
    tx, ty = _aa_weights(W0, nw, cubic), _aa_weights(H0, nh, cubic)
    hpass = np.empty((C, H0, nw), F32)
    for j, (x0, w) in enumerate(tx):
        acc = np.zeros((C, H0), F32)
        for k in range(len(w)):
            acc += w[k] * x[:, :, x0 + k]
        hpass[:, :, j] = acc
    out = np.empty((C, nh, nw), F32)
    for i, (y0, w) in enumerate(ty):
        acc = np.zeros((C, nw), F32)
        for k in range(len(w)):
            acc += w[k] * hpass[:, y0 + k, :]
        out[:, i, :] = acc
    return out


def process_frame(img_bgr: np.ndarray, target_height: int) -> np.ndarray:
    """reference depth.py:540-566 (torch branch): HWC uint8 BGR(A) -> CHW float32 RGB 0..255; if target_height < H0,
    F.interpolate(bilinear, align_corners=False, antialias=True) to ((t//2)*2, (int(W0*t/H0)//2)*2): separable,
    horizontal pass then vertical pass."""
    x = np.ascontiguousarray(img_bgr[..., :3][..., ::-1].transpose(2, 0, 1)).astype(F32)
    _, H0, W0 = x.shape
    if target_height >= H0:
        return x
    nh = (target_height // 2) * 2
    nw = (int(W0 * target_height / H0) // 2) * 2
    return separable_aa_resize(x, nh, nw, cubic=False)


def process_tensor(img: np.ndarray, height: int) -> np.ndarray:
    """reference depth.py:576-601: the tensor branch of the NON-CUDA process() ("tensor capture path is already RGB CHW"):
    [3|4,H,W] -> first three planes, or [H,W,>=3] -> [..., :3] as CHW; NO channel flip; height >= H0 returns the frame as is;
    else F.interpolate(bilinear, align_corners=False, antialias=False) to ((height//2)*2, (int(W0*height/H0)//2)*2)."""
    img = np.asarray(img)
    if img.ndim == 3 and img.shape[0] in (3, 4):
        x = img[:3]
    elif img.ndim == 3 and img.shape[-1] >= 3:
        x = np.ascontiguousarray(img[..., :3].transpose(2, 0, 1))
    else:
        raise ValueError(f"Unsupported tensor image shape: {tuple(img.shape)}")
    _, h0, w0 = x.shape
    if height >= h0:
        return x
    width = (int(w0 * height / h0) // 2) * 2
    height = (height // 2) * 2
    return bilinear_resize(x.astype(F32), height, width, align_corners=False)


def _area_tab(ssize: int, dsize: int):
    """OpenCV computeResizeAreaTab (modules/imgproc/src/resize.cpp; opencv-python 4.12.0.88 is the reference's pin,
    requirements.txt:4 -- third party, not under /root/reference, cv2 is not installed in this image): the source cells
    [sx] and float32 weights each destination index dx accumulates when scale = ssize / dsize (double) is not an integer."""
    scale = 1.0 / (float(dsize) / float(ssize))          # resize(): inv_scale = dsize / ssize, scale = 1. / inv_scale (not ssize / dsize: 1 ulp apart)
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, F32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, F32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, F32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area_u8(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_AREA) for a DOWN-scale of uint8 [H,W,C], restated from OpenCV's
    published algorithm (resize.cpp): integer scale factors -> ResizeAreaFast (2x2: (a+b+c+d+2)>>2; else
    saturate_cast<uchar>(sum * (1.f/area)), cvRound = round-half-even); otherwise ResizeArea_<uchar,float> -- per source row a
    horizontal accumulation buf[dx] += S[sx]*alpha in table order, then sum[dx] (+)= beta*buf[dx] over the rows of a
    destination row, saturate_cast<uchar> at its end.  PARITY UNPINNED against cv2 itself (not installed here): held to an
    exact float64 area integral within 1 level in tests/test_oracle_golden.py."""
    img = np.asarray(img)
    H, W, C = img.shape
    sx, sy = 1.0 / (dw / float(W)), 1.0 / (dh / float(H))      # as cv::resize forms them (inv_scale first)
    ix, iy = int(round(sx)), int(round(sy))
    if abs(sx - ix) < np.finfo(np.float64).eps and abs(sy - iy) < np.finfo(np.float64).eps:
        blk = img[: dh * iy, : dw * ix].reshape(dh, iy, dw, ix, C).astype(np.int64).sum(axis=(1, 3))
        if ix == 2 and iy == 2:
            return ((blk + 2) >> 2).astype(np.uint8)
        return np.clip(np.rint(blk.astype(F32) * F32(1.0 / (ix * iy))), 0, 255).astype(np.uint8)
    xt, yt = _area_tab(W, dw), _area_tab(H, dh)
    xd = np.array([t[0] for t in xt]); xs = np.array([t[1] for t in xt]); xa = np.array([t[2] for t in xt], F32)
    src = img.astype(F32)
    out = np.empty((dh, dw, C), np.uint8)
    acc = None
    prev = yt[0][0]
    for dy, syi, beta in yt:
        buf = np.zeros((dw, C), F32)
        row = src[syi]
        for k in range(len(xd)):                                 # table order (np.add.at would reorder nothing, but is slower)
            buf[xd[k]] = buf[xd[k]] + row[xs[k]] * xa[k]
        if dy != prev:
            out[prev] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
            acc = beta * buf
            prev = dy
        elif acc is None:
            acc = beta * buf
        else:
            acc = acc + beta * buf
    out[prev] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return out


def process_area(img_bgr: np.ndarray, height: int) -> np.ndarray:
    """reference depth.py:603-629: the numpy branch of the NON-CUDA process(): cv2.cvtColor BGR(A) -> RGB, then, only if
    height < H0, cv2.resize to (int(W0*height/H0), height) with INTER_AREA (no rounding to even sizes here); uint8 HWC."""
    h0, w0 = img_bgr.shape[:2]
    width = int(w0 * height / h0)
    rgb = np.ascontiguousarray(img_bgr[..., :3][..., ::-1])
    if height >= h0:
        return rgb
    return resize_area_u8(rgb, width, height)


_FONT = {  # reference depth.py:641-658 (5x3 glyphs)
    "0": "111101101101111", "1": "010110010010111", "2": "111001111100111", "3": "111001111001111",
    "4": "101101111001001", "5": "111100111001111", "6": "111100111101111", "7": "111001010100100",
    "8": "111101111101111", "9": "111101111001111", "F": "111100110100100", "P": "110101110100100",
    "S": "111100111001111", ":": "000010000010000", ".": "000000000000010", " ": "000000000000000",
}


def overlay_text(rgb_chw: np.ndarray, text: str) -> np.ndarray:
    """reference depth.py:2061-2103 for one freshly built mask: glyphs scaled by max(1,min(8,H//60)), margin 2*scale,
    spacing scale, clipped to the frame; rgb*(1-mask) + (0,255,0)*mask."""
    _, H, W = rgb_chw.shape
    scale = max(1, min(8, H // 60))
    cw, ch = 3 * scale, 5 * scale
    mask = np.zeros((H, W), F32)
    for i, c in enumerate(text):
        g = np.array([int(b) for b in _FONT.get(c, _FONT[" "])], F32).reshape(5, 3)
        g = np.repeat(np.repeat(g, scale, 0), scale, 1)
        x0, y0 = 2 * scale + i * (cw + scale), 2 * scale
        x1, y1 = min(W, x0 + cw), min(H, y0 + ch)
        if x0 < W and y0 < H:
            mask[y0:y1, x0:x1] = np.maximum(mask[y0:y1, x0:x1], g[:y1 - y0, :x1 - x0])
    color = np.array([0, 255, 0], F32).reshape(3, 1, 1)
    return (rgb_chw.astype(F32) * (1 - mask) + color * mask).astype(F32)


# ----------------------------------------------------------------------------------------------
# whole path
# ----------------------------------------------------------------------------------------------
class PipelineOracle:
    """predict_depth + make_sbs of the reference, CPU branch, float32 (autocast disabled)."""

    def __init__(self, cfg, weights, depth_resolution=518, foreground_scale=0.05, aa_strength=4.0,
                 ema_alpha=0.9, metric=False, max_depth=0.0, cuda_branch=False, square=False):
        self.model = DepthAnythingOracle(cfg, weights, max_depth)
        self.cuda_branch = cuda_branch              # _resize_patch_aligned_t's IS_CUDA branch (bicubic + antialias)
        self.square = square                        # get_patch_size() is None (CAPTURE_MODE "Window"): fixed-square input
        self.metric = metric
        self.target = depth_resolution
        self.fg = foreground_scale
        self.aa = aa_strength
        self.stab = DepthStabilizer(ema_alpha)

    def model_input(self, img_hwc_u8: np.ndarray) -> np.ndarray:
        if self.square:
            return normalise(resize_fixed_square(np.ascontiguousarray(img_hwc_u8.transpose(2, 0, 1)), self.target))
        x = resize_patch_aligned(np.ascontiguousarray(img_hwc_u8.transpose(2, 0, 1)), self.target,
                                 self.model.cfg.patch, self.cuda_branch)
        return normalise(x)

    def predict_depth(self, img_hwc_u8: np.ndarray, use_temporal_smooth=False, taps=None) -> np.ndarray:
        """reference depth.py:1897-2025."""
        H, W = img_hwc_u8.shape[:2]
        x = self.model_input(img_hwc_u8)
        raw = self.model.forward(x, taps)
        d = post_process_depth(raw, self.fg, self.aa, metric=self.metric)
        if taps is not None:
            taps["model_input"] = x
            taps["raw_depth"] = raw
            taps["post_depth"] = d
        if use_temporal_smooth:
            d = self.stab(d)
        return upsample_depth(d, H, W)

    def make_sbs(self, img_hwc_u8, depth, **kw) -> np.ndarray:
        """reference depth.py:2186-2231 -> HWC float32 0..255."""
        rgb = img_hwc_u8.transpose(2, 0, 1).astype(F32)
        return np.ascontiguousarray(make_sbs_core(rgb, depth, **kw).transpose(1, 2, 0))
