"""Import the REFERENCE's depth.py in this container (never on the GPU box).

Follows SURVEY.md Appendix A: a scratch cwd with an edited copy of settings.yaml, stub
modules for cv2 / easydict (absent here), and ``AutoModelForDepthEstimation.from_pretrained``
patched to build a random-init HF ``DepthAnythingForDepthEstimation`` of the requested
architecture filled with THIS repo's deterministic weights (desktop2stereo_amd.weights).

Only tests/golden/make_golden.py and ad-hoc validation use this file. It reads
/root/reference and therefore must not be imported by ``-m gpu`` tests, smoke() or bench.py.
"""
from __future__ import annotations

import contextlib
import os
import sys
import tempfile
import types

REF = "/root/reference"

_HF_NAME = {"vits": "Depth-Anything-V2-Small", "vitb": "Depth-Anything-V2-Base",
            "vitl": "Depth-Anything-V2-Large", "tiny": "Depth-Anything-V2-Small"}


def load_reference(model: str = "tiny", depth_resolution: int = 518, seed: int = 0,
                   fp32: bool = True, aa: int = 2, fg: float = 0.5, metric: str = ""):
    """Returns the reference ``depth`` module, model built with our weights.

    fp32=True patches depth.maybe_autocast -> nullcontext (reference depth.py:661-664), giving
    the fp32 oracle; fp32=False leaves the as-shipped CPU autocast (bf16).
    One reference import per process (module-level model construction, depth.py:1784).
    """
    import yaml
    import torch
    import transformers
    from transformers import DepthAnythingConfig, DepthAnythingForDepthEstimation

    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.weights import make_weights

    scratch = tempfile.mkdtemp(prefix="d2s_ref_")
    with open(os.path.join(REF, "settings.yaml")) as f:
        st = yaml.safe_load(f)
    # metric = "Indoor" | "Outdoor": the reference's Depth-Anything-V2-Metric-* ids (utils.py:761-769): is_metric()
    # becomes true (depth.py:666) and the HF head is built with depth_estimation_type="metric"
    name = _HF_NAME[model].replace("V2-", f"V2-Metric-{metric}-") if metric else _HF_NAME[model]
    max_depth = {"": None, "Indoor": 20, "Outdoor": 80}[metric]
    st.update({"Depth Model": name, "FP16": False, "Depth Resolution": depth_resolution,
               "Run Mode": "Legacy Streamer", "torch.compile": False, "TensorRT": False,
               "CoreML": False, "OpenVINO": False, "MIGraphX": False,
               "Anti-aliasing": aa, "Foreground Scale": fg})
    with open(os.path.join(scratch, "settings.yaml"), "w") as f:
        yaml.safe_dump(st, f)
    os.chdir(scratch)
    sys.path.insert(0, REF)

    cv2 = types.ModuleType("cv2")
    cv2.UMat = type("UMat", (), {})
    sys.modules["cv2"] = cv2                                   # depth.py:17, 570
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.__dict__ = self
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed                                # dpt_temporal.py:19

    cfg = MODELS[model]

    def fake_from_pretrained(model_id, **kw):                   # replaces depth.py:1649-1662
        hf = DepthAnythingConfig(
            backbone_config=dict(model_type="dinov2", hidden_size=cfg.hidden,
                                 num_attention_heads=cfg.heads, num_hidden_layers=cfg.layers,
                                 image_size=518, patch_size=14, out_indices=list(cfg.out_indices),
                                 apply_layernorm=True, reshape_hidden_states=False),
            reassemble_hidden_size=cfg.hidden, neck_hidden_sizes=list(cfg.neck),
            fusion_hidden_size=cfg.fusion, head_hidden_size=cfg.head_hidden,
            **({"depth_estimation_type": "metric", "max_depth": max_depth} if metric else {}))
        m = DepthAnythingForDepthEstimation(hf)
        sd = {k: torch.from_numpy(v) for k, v in make_weights(cfg, seed).items()}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        missing = [k for k in missing if "mask_token" not in k]
        assert not missing and not unexpected, (missing, unexpected)
        return m.eval()

    transformers.AutoModelForDepthEstimation.from_pretrained = staticmethod(fake_from_pretrained)
    import depth as D                                            # builds D.model_wraper at import
    if fp32:
        D.maybe_autocast = lambda *a, **k: contextlib.nullcontext()
    return D
