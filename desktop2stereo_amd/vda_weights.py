"""Video-Depth-Anything (streaming) weights: temporal-module tensors, key mapping to/from the
reference's checkpoint layout, and the seeded synthetic generator.

The backbone and DPT head of VDA are the same arithmetic as Depth-Anything-v2 (reference
models/video_depth_anything/dinov2.py, dpt.py, util/blocks.py), so the engine and oracle keep the HF
key names for them; only the four temporal modules (reference dpt_temporal.py:50-60,
motion_module/motion_module.py) add tensors, kept under the reference's own names
``head.motion_modules.{m}.temporal_transformer.*``.  ``vda_to_hf`` converts a reference ``.pth``
state dict (``pretrained.*`` / ``head.*``, reference depth.py:889-902) for the engine;
``hf_to_vda`` is its inverse (used to fill the reference model when generating golden vectors).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import numpy as np

from .config import ModelConfig
from .weights import make_weights

N_MODULES = 4
N_ATTN = 2             # num_attention_blocks
T_HEADS = 8            # num_attention_heads
T_WINDOW = 32          # INFER_LEN (reference vda2_s.py:29)


def temporal_channels(cfg: ModelConfig):
    """in_channels of the 4 TemporalModules: layer_3, layer_4, path_4, path_3 (dpt_temporal.py:50-60)."""
    return (cfg.neck[2], cfg.neck[3], cfg.fusion, cfg.fusion)


def temporal_shapes(cfg: ModelConfig) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    for m, C in enumerate(temporal_channels(cfg)):
        p = f"head.motion_modules.{m}.temporal_transformer."
        s[p + "norm.weight"] = (C,)
        s[p + "norm.bias"] = (C,)
        s[p + "proj_in.weight"] = (C, C)
        s[p + "proj_in.bias"] = (C,)
        b = p + "transformer_blocks.0."
        for a in range(N_ATTN):
            q = b + f"attention_blocks.{a}."
            s[q + "to_q.weight"] = (C, C)
            s[q + "to_k.weight"] = (C, C)
            s[q + "to_v.weight"] = (C, C)
            s[q + "to_out.0.weight"] = (C, C)
            s[q + "to_out.0.bias"] = (C,)
            s[b + f"norms.{a}.weight"] = (C,)
            s[b + f"norms.{a}.bias"] = (C,)
        s[b + "ff.net.0.proj.weight"] = (8 * C, C)
        s[b + "ff.net.0.proj.bias"] = (8 * C,)
        s[b + "ff.net.2.weight"] = (C, 4 * C)
        s[b + "ff.net.2.bias"] = (C,)
        s[b + "ff_norm.weight"] = (C,)
        s[b + "ff_norm.bias"] = (C,)
        s[p + "proj_out.weight"] = (C, C)
        s[p + "proj_out.bias"] = (C,)
    return s


def positional_encoding(C: int, max_len: int = T_WINDOW) -> np.ndarray:
    """Sinusoidal table of PositionalEncoding (reference motion_module.py:214-222), float32 like torch."""
    import torch
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, C, 2) * (-math.log(10000.0) / C))
    pe = torch.zeros(max_len, C)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.numpy()


def make_vda_weights(cfg: ModelConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """HF-keyed backbone/DPT weights (make_weights) + seeded temporal tensors.  proj_out is NOT zero
    (the reference zero-initialises it before training, motion_module.py:60-61) so that the modules matter."""
    out = make_weights(cfg, seed)
    for name, shape in temporal_shapes(cfg).items():
        g = np.random.default_rng([seed, zlib.crc32(name.encode())])
        leaf = name.rsplit(".", 1)[-1]
        if any(t in name for t in (".norm.", ".norms.", ".ff_norm.")):
            w = 1.0 + g.normal(0.0, 0.1, shape) if leaf == "weight" else g.normal(0.0, 0.1, shape)
        elif leaf == "bias":
            w = g.normal(0.0, 0.05, shape)
        else:
            w = g.normal(0.0, 1.0 / np.sqrt(shape[1]), shape)
            if "proj_out" in name:
                w *= 0.5
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


# ---- key mapping ---------------------------------------------------------------------------------
def _pairs(cfg: ModelConfig):
    """(hf_name, vda_name) for every 1:1 tensor."""
    P = [("backbone.embeddings.cls_token", "pretrained.cls_token"),
         ("backbone.embeddings.position_embeddings", "pretrained.pos_embed"),
         ("backbone.embeddings.patch_embeddings.projection.weight", "pretrained.patch_embed.proj.weight"),
         ("backbone.embeddings.patch_embeddings.projection.bias", "pretrained.patch_embed.proj.bias"),
         ("backbone.layernorm.weight", "pretrained.norm.weight"), ("backbone.layernorm.bias", "pretrained.norm.bias")]
    for i in range(cfg.layers):
        h, v = f"backbone.encoder.layer.{i}.", f"pretrained.blocks.{i}."
        for n in ("norm1", "norm2"):
            P += [(h + n + ".weight", v + n + ".weight"), (h + n + ".bias", v + n + ".bias")]
        P += [(h + "attention.output.dense.weight", v + "attn.proj.weight"), (h + "attention.output.dense.bias", v + "attn.proj.bias"),
              (h + "layer_scale1.lambda1", v + "ls1.gamma"), (h + "layer_scale2.lambda1", v + "ls2.gamma")]
        for n in ("fc1", "fc2"):
            P += [(h + f"mlp.{n}.weight", v + f"mlp.{n}.weight"), (h + f"mlp.{n}.bias", v + f"mlp.{n}.bias")]
    for i in range(4):
        h = f"neck.reassemble_stage.layers.{i}."
        P += [(h + "projection.weight", f"head.projects.{i}.weight"), (h + "projection.bias", f"head.projects.{i}.bias")]
        if i != 2:
            P += [(h + "resize.weight", f"head.resize_layers.{i}.weight"), (h + "resize.bias", f"head.resize_layers.{i}.bias")]
        P += [(f"neck.convs.{i}.weight", f"head.scratch.layer{i + 1}_rn.weight")]
    for j in range(4):                                    # HF fusion layer j = refinenet(4-j)
        h, v = f"neck.fusion_stage.layers.{j}.", f"head.scratch.refinenet{4 - j}."
        P += [(h + "projection.weight", v + "out_conv.weight"), (h + "projection.bias", v + "out_conv.bias")]
        for r in (1, 2):
            for c in (1, 2):
                for leaf in ("weight", "bias"):
                    P += [(h + f"residual_layer{r}.convolution{c}.{leaf}", v + f"resConfUnit{r}.conv{c}.{leaf}")]
    P += [("head.conv1.weight", "head.scratch.output_conv1.weight"), ("head.conv1.bias", "head.scratch.output_conv1.bias"),
          ("head.conv2.weight", "head.scratch.output_conv2.0.weight"), ("head.conv2.bias", "head.scratch.output_conv2.0.bias"),
          ("head.conv3.weight", "head.scratch.output_conv2.2.weight"), ("head.conv3.bias", "head.scratch.output_conv2.2.bias")]
    return P


def hf_to_vda(sd: Dict[str, np.ndarray], cfg: ModelConfig) -> Dict[str, np.ndarray]:
    out = {v: sd[h] for h, v in _pairs(cfg)}
    for i in range(cfg.layers):
        h, v = f"backbone.encoder.layer.{i}.attention.attention.", f"pretrained.blocks.{i}.attn.qkv."
        out[v + "weight"] = np.concatenate([sd[h + f"{n}.weight"] for n in ("query", "key", "value")], 0)
        out[v + "bias"] = np.concatenate([sd[h + f"{n}.bias"] for n in ("query", "key", "value")], 0)
    out["pretrained.mask_token"] = np.zeros((1, cfg.hidden), np.float32)
    for k, a in sd.items():
        if k.startswith("head.motion_modules."):
            out[k] = a
    for m, C in enumerate(temporal_channels(cfg)):
        for a in range(N_ATTN):
            out[f"head.motion_modules.{m}.temporal_transformer.transformer_blocks.0.attention_blocks.{a}.pos_encoder.pe"] = \
                positional_encoding(C)[None]
    return out


def vda_to_hf(sd: Dict[str, np.ndarray], cfg: ModelConfig) -> Dict[str, np.ndarray]:
    out = {h: np.asarray(sd[v], dtype=np.float32) for h, v in _pairs(cfg)}
    D = cfg.hidden
    for i in range(cfg.layers):
        h, v = f"backbone.encoder.layer.{i}.attention.attention.", f"pretrained.blocks.{i}.attn.qkv."
        w, b = np.asarray(sd[v + "weight"], np.float32), np.asarray(sd[v + "bias"], np.float32)
        for j, n in enumerate(("query", "key", "value")):
            out[h + f"{n}.weight"] = np.ascontiguousarray(w[j * D:(j + 1) * D])
            out[h + f"{n}.bias"] = np.ascontiguousarray(b[j * D:(j + 1) * D])
    for k, a in sd.items():
        if k.startswith("head.motion_modules.") and not k.endswith("pos_encoder.pe"):
            out[k] = np.asarray(a, dtype=np.float32)
    return out
