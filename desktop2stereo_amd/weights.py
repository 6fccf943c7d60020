"""Depth-Anything-v2 weights: HF key layout, deterministic synthetic generator, loader.

There is no network and no checkpoint on disk, so benches and parity tests run on
seeded synthetic weights of the exact HF ``DepthAnythingForDepthEstimation`` layout
(key names / shapes measured in SURVEY.md section 8c, "Weight layout contract").
Real ``model.safetensors`` files (reference convert.py:14-24 writes them) load through
``load_safetensors`` with the same keys.

The generator is the build's own: one numpy Generator per tensor, keyed by
(seed, crc32(name)), so any subset can be regenerated independently, on any host.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

from .config import ModelConfig

REASSEMBLE_FACTORS = (4, 2, 1, 0.5)   # HF DepthAnythingConfig default


def expected_shapes(cfg: ModelConfig) -> Dict[str, tuple]:
    """name -> shape for every tensor the engine consumes (HF key names)."""
    D, F = cfg.hidden, cfg.fusion
    s: Dict[str, tuple] = {}
    e = "backbone.embeddings."
    s[e + "cls_token"] = (1, 1, D)
    s[e + "position_embeddings"] = (1, cfg.pos_grid * cfg.pos_grid + 1, D)
    s[e + "patch_embeddings.projection.weight"] = (D, 3, cfg.patch, cfg.patch)
    s[e + "patch_embeddings.projection.bias"] = (D,)
    for i in range(cfg.layers):
        p = f"backbone.encoder.layer.{i}."
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"] = (D,)
            s[p + n + ".bias"] = (D,)
        for n in ("query", "key", "value"):
            s[p + f"attention.attention.{n}.weight"] = (D, D)
            s[p + f"attention.attention.{n}.bias"] = (D,)
        s[p + "attention.output.dense.weight"] = (D, D)
        s[p + "attention.output.dense.bias"] = (D,)
        s[p + "layer_scale1.lambda1"] = (D,)
        s[p + "layer_scale2.lambda1"] = (D,)
        s[p + "mlp.fc1.weight"] = (cfg.mlp, D)
        s[p + "mlp.fc1.bias"] = (cfg.mlp,)
        s[p + "mlp.fc2.weight"] = (D, cfg.mlp)
        s[p + "mlp.fc2.bias"] = (D,)
    s["backbone.layernorm.weight"] = (D,)
    s["backbone.layernorm.bias"] = (D,)
    for i, (c, f) in enumerate(zip(cfg.neck, REASSEMBLE_FACTORS)):
        p = f"neck.reassemble_stage.layers.{i}."
        s[p + "projection.weight"] = (c, D, 1, 1)
        s[p + "projection.bias"] = (c,)
        if f > 1:
            s[p + "resize.weight"] = (c, c, int(f), int(f))      # ConvTranspose2d [in,out,k,k]
            s[p + "resize.bias"] = (c,)
        elif f < 1:
            s[p + "resize.weight"] = (c, c, 3, 3)                # Conv2d stride 2 pad 1
            s[p + "resize.bias"] = (c,)
        s[f"neck.convs.{i}.weight"] = (F, c, 3, 3)               # bias=False
    for i in range(4):
        p = f"neck.fusion_stage.layers.{i}."
        s[p + "projection.weight"] = (F, F, 1, 1)
        s[p + "projection.bias"] = (F,)
        for r in ("residual_layer1", "residual_layer2"):
            for c in ("convolution1", "convolution2"):
                s[p + f"{r}.{c}.weight"] = (F, F, 3, 3)
                s[p + f"{r}.{c}.bias"] = (F,)
    s["head.conv1.weight"] = (F // 2, F, 3, 3)
    s["head.conv1.bias"] = (F // 2,)
    s["head.conv2.weight"] = (cfg.head_hidden, F // 2, 3, 3)
    s["head.conv2.bias"] = (cfg.head_hidden,)
    s["head.conv3.weight"] = (1, cfg.head_hidden, 1, 1)
    s["head.conv3.bias"] = (1,)
    return s


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def make_weights(cfg: ModelConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded synthetic state dict (float32) with HF key names.

    Scales are chosen so activations stay O(1) through the network and the predicted
    depth has a healthy dynamic range (percentile min-max normalisation downstream,
    reference depth.py:816-867, amplifies error when the raw field is nearly flat):
    fan-in-scaled normal matrices, LN gamma ~ 1 +- 0.1, LayerScale ~ U(0.3, 0.7),
    small non-zero biases so no ReLU dead-zones.
    """
    out: Dict[str, np.ndarray] = {}
    for name, shape in expected_shapes(cfg).items():
        g = _rng(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("cls_token"):
            w = g.normal(0.0, 0.5, shape)
        elif name.endswith("position_embeddings"):
            w = g.normal(0.0, 0.5, shape)
        elif "lambda1" in name:
            w = g.uniform(0.3, 0.7, shape)
        elif ".norm" in name or "layernorm" in name:
            w = 1.0 + g.normal(0.0, 0.1, shape) if leaf == "weight" else g.normal(0.0, 0.1, shape)
        elif leaf == "bias":
            w = g.normal(0.0, 0.05, shape)
            if name == "head.conv3.bias":
                w = np.full(shape, 0.25)
        else:  # matrices / conv kernels
            if "resize.weight" in name and len(shape) == 4 and shape[2] in (2, 4) and "reassemble" in name:
                fan_in = shape[0]                      # ConvTranspose k==s: one tap per output pixel
            else:
                fan_in = int(np.prod(shape[1:]))
            gain = 1.0
            if name.startswith("neck.") or name.startswith("head."):
                gain = 1.4                              # ReLU-preceded convs
            if name == "head.conv3.weight":
                gain = 2.0
            w = g.normal(0.0, gain / np.sqrt(fan_in), shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def load_safetensors(path: str, cfg: ModelConfig) -> Dict[str, np.ndarray]:
    """Load a HF ``model.safetensors`` (fp16/bf16/fp32) into float32 numpy, checking shapes."""
    from safetensors import safe_open
    import torch
    out: Dict[str, np.ndarray] = {}
    want = expected_shapes(cfg)
    with safe_open(path, framework="pt") as f:
        keys = set(f.keys())
        for name, shape in want.items():
            if name not in keys:
                raise KeyError(f"{path}: missing tensor {name}")
            t = f.get_tensor(name).to(torch.float32)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name}: shape {tuple(t.shape)} != expected {shape}")
            out[name] = t.contiguous().numpy()
    return out
