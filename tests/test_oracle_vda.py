"""CPU: the numpy VDA oracle (oracle/vda_oracle.py) against goldens captured from the reference's own
streaming VideoDepthAnything (tests/golden/make_golden_vda.py): 5 frames in order, depth per frame and the
8 temporal caches after the last frame."""
import json
import os

import numpy as np

from desktop2stereo_amd.config import MODELS
from desktop2stereo_amd.vda_weights import hf_to_vda, make_vda_weights, vda_to_hf
from oracle.vda_oracle import VideoDepthOracle


def test_key_mapping_round_trip():
    cfg = MODELS["tiny"]
    w = make_vda_weights(cfg, 0)
    back = vda_to_hf(hf_to_vda(w, cfg), cfg)
    assert set(back) == set(w)
    assert all(np.array_equal(back[k], w[k]) for k in w)


def test_vda_tiny_stream(golden_dir):
    cfg = MODELS["tiny"]
    z = np.load(os.path.join(golden_dir, "vda_tiny.npz"))
    meta = json.load(open(os.path.join(golden_dir, "vda_tiny.json")))
    orc = VideoDepthOracle(cfg, make_vda_weights(cfg, 0))
    for fi in range(len(meta["frames"])):
        d = orc.forward(z[f"f{fi}_x"])
        ref = z[f"f{fi}_depth"]
        assert np.abs(d - ref).max() <= 3e-5 * max(1.0, float(ref.max())), (fi, np.abs(d - ref).max(), ref.max())
    ci = 0
    for m in orc.modules:
        for a in range(2):
            assert np.abs(m.cache[a] - z[f"cache{ci}"]).max() <= 5e-5, ci
            ci += 1


def test_vda_tiny_window_wrap(golden_dir):
    """40 frames from the reference (> the 32-frame window): update_cache's shift order (vda2_s.py:177-187) as the
    reference executes it pins the oracle's eviction order beyond frame 32."""
    from desktop2stereo_amd import synth
    from oracle import d2s_oracle as O
    cfg = MODELS["tiny"]
    z = np.load(os.path.join(golden_dir, "vda_tiny_long.npz"))
    meta = json.load(open(os.path.join(golden_dir, "vda_tiny_long.json")))
    assert len(meta["frames"]) >= 34
    orc = VideoDepthOracle(cfg, make_vda_weights(cfg, 0))
    for fi, fr in enumerate(meta["frames"]):
        frame = synth.structured_frame(fr["h"], fr["w"], fr["seed"])
        x = O.normalise(O.resize_patch_aligned(np.ascontiguousarray(frame.transpose(2, 0, 1)), meta["depth_resolution"]))
        d = orc.forward(x)
        ref = z[f"f{fi}_depth"]
        assert np.abs(d - ref).max() <= 5e-5 * max(1.0, float(ref.max())), (fi, np.abs(d - ref).max(), ref.max())
