"""CPU oracle for streaming Video-Depth-Anything  --  TEST INFRASTRUCTURE, NOT PRODUCT.

numpy float32 restatement of the reference's in-tree VDA (A17 in SURVEY.md section 8a):
    VideoDepthAnything.forward / update_cache        models/video_depth_anything/vda2_s.py:177-224
    DPTHeadTemporal.forward                           models/video_depth_anything/dpt_temporal.py:62-138
    TemporalTransformer3DModel / TemporalTransformerBlock / TemporalAttention
                                                      motion_module/motion_module.py:102-134, 164-196, 242-321
    CrossAttention._attention, FeedForward (GEGLU)    motion_module/attention.py:182-211, 296-384
The backbone + DPT head are DepthAnythingOracle's arithmetic (same structure, reference dpt.py,
util/blocks.py); only the position-embedding resample differs (interpolate_offset 0.1,
dinov2.py:179-210).  Pinned by tests/golden/vda_*.npz, captured by running the reference's own
VideoDepthAnything in the build container (tests/golden/make_golden_vda.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from .d2s_oracle import F32, DepthAnythingOracle, gelu_erf, layer_norm

N_ATTN, T_HEADS, T_WINDOW = 2, 8, 32


def positional_encoding(C: int, max_len: int = T_WINDOW) -> np.ndarray:
    """motion_module.py:214-222 (float32, computed like torch does)."""
    import torch
    import math
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, C, 2) * (-math.log(10000.0) / C))
    pe = torch.zeros(max_len, C)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.numpy()


def group_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, groups: int = 32, eps: float = 1e-6) -> np.ndarray:
    """nn.GroupNorm on [C,h,w]."""
    C, h, w = x.shape
    xg = x.reshape(groups, -1)
    mu = xg.mean(1, keepdims=True, dtype=F32)
    var = ((xg - mu) ** 2).mean(1, keepdims=True, dtype=F32)
    y = ((xg - mu) / np.sqrt(var + F32(eps))).reshape(C, h, w)
    return (y * g.reshape(C, 1, 1) + b.reshape(C, 1, 1)).astype(F32)


class TemporalModuleOracle:
    """One TemporalModule in streaming mode (T = 1, cache of 31 normed hidden states per attention block)."""

    def __init__(self, w: Dict[str, np.ndarray], m: int, C: int):
        self.p = f"head.motion_modules.{m}.temporal_transformer."
        self.w = w
        self.C = C
        self.pe = positional_encoding(C)
        self.cache: Optional[List[np.ndarray]] = None        # N_ATTN arrays [sites, 31, C]

    def reset(self):
        self.cache = None

    def _attn(self, a: int, n: np.ndarray) -> np.ndarray:
        """TemporalAttention.forward (motion_module.py:268-321): q from the current frame only."""
        w, C = self.w, self.C
        q_ = self.p + f"transformer_blocks.0.attention_blocks.{a}."
        sites = n.shape[0]
        cur = n[:, None, :]                                             # [(b d), f=1, c]
        hid = cur if self.cache is None else np.concatenate([self.cache[a], cur], 1)
        d_in = hid.shape[1] - 1
        hid = hid + self.pe[None, :hid.shape[1]]                        # pos_encoder (APE by index in the window)
        q = hid[:, d_in:] @ w[q_ + "to_q.weight"].T                     # no bias
        k = hid @ w[q_ + "to_k.weight"].T
        v = hid @ w[q_ + "to_v.weight"].T
        dh = C // T_HEADS
        q = q.reshape(sites, 1, T_HEADS, dh).transpose(0, 2, 1, 3)
        k = k.reshape(sites, -1, T_HEADS, dh).transpose(0, 2, 1, 3)
        v = v.reshape(sites, -1, T_HEADS, dh).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) * F32(dh ** -0.5)
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        pr = e / e.sum(-1, keepdims=True, dtype=F32)
        o = (pr @ v).transpose(0, 2, 1, 3).reshape(sites, C)
        return (o @ w[q_ + "to_out.0.weight"].T + w[q_ + "to_out.0.bias"]).astype(F32)

    def __call__(self, x: np.ndarray) -> np.ndarray:
        """x [C,h,w] -> [C,h,w]; updates the per-block caches (vda2_s.py:177-187, 203-218)."""
        w, C, p = self.w, self.C, self.p
        _, h, wd = x.shape
        hs = group_norm(x, w[p + "norm.weight"], w[p + "norm.bias"], 32, 1e-6)
        hs = hs.reshape(C, h * wd).T @ w[p + "proj_in.weight"].T + w[p + "proj_in.bias"]
        b = p + "transformer_blocks.0."
        new_states = []
        for a in range(N_ATTN):
            n = layer_norm(hs, w[b + f"norms.{a}.weight"], w[b + f"norms.{a}.bias"], 1e-5)
            hs = self._attn(a, n) + hs
            new_states.append(n)
        n = layer_norm(hs, w[b + "ff_norm.weight"], w[b + "ff_norm.bias"], 1e-5)
        u = n @ w[b + "ff.net.0.proj.weight"].T + w[b + "ff.net.0.proj.bias"]
        u = u[:, :4 * C] * gelu_erf(u[:, 4 * C:])                       # GEGLU
        hs = (u @ w[b + "ff.net.2.weight"].T + w[b + "ff.net.2.bias"] + hs).astype(F32)
        out = hs @ w[p + "proj_out.weight"].T + w[p + "proj_out.bias"]
        out = out.T.reshape(C, h, wd) + x
        if self.cache is None:                                          # first frame: 31 copies (vda2_s.py:203-207)
            self.cache = [np.repeat(s[:, None, :], T_WINDOW - 1, 1) for s in new_states]
        else:                                                           # drop oldest, append newest
            self.cache = [np.concatenate([c[:, 1:], s[:, None, :]], 1) for c, s in zip(self.cache, new_states)]
        return out.astype(F32)


class VideoDepthOracle(DepthAnythingOracle):
    """Streaming VideoDepthAnything.forward: one frame in, depth [h,w] out, stateful."""
    pos_offset = 0.1                                                    # vendored DINOv2 interpolate_offset

    def __init__(self, cfg, weights):
        super().__init__(cfg, weights)
        from desktop2stereo_amd.vda_weights import temporal_channels
        self.modules = [TemporalModuleOracle(self.w, m, C) for m, C in enumerate(temporal_channels(cfg))]

    def reset(self):
        for m in self.modules:
            m.reset()

    def forward(self, x: np.ndarray, taps: Optional[dict] = None) -> np.ndarray:
        hooks = {"layer_3": self.modules[0], "layer_4": self.modules[1], "path_4": self.modules[2], "path_3": self.modules[3]}
        # forward_depth ends with relu(interpolate(out, (H,W), align_corners=True)) at the same size: identity on a ReLU output
        return super().forward(x, taps, hooks)
