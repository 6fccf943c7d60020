"""GPU: streaming Video-Depth-Anything engine (A17) against goldens captured from the reference's own
VideoDepthAnything (tests/golden/make_golden_vda.py) and against the numpy oracle.
fp32 engine: every frame of the stream within 2e-4 of the range; bf16 engine: bf16-class tolerance."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no ROCm device is visible")
    return torch.device("cuda", 0)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-4), ("bf16", 0.036)])
def test_vda_tiny_stream(dev, golden_dir, prec, tol):
    from desktop2stereo_amd import ops
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.vda_weights import make_vda_weights
    cfg = MODELS["tiny"]
    z = np.load(os.path.join(golden_dir, "vda_tiny.npz"))
    meta = json.load(open(os.path.join(golden_dir, "vda_tiny.json")))
    eng = ops.Engine(cfg, make_vda_weights(cfg, 0), 42, 84, 1, prec, temporal=True)
    for rep in range(2):                                      # second pass after reset_stream must reproduce the first
        eng.reset_stream()
        for fi in range(len(meta["frames"])):
            d = eng(_t(z[f"f{fi}_x"], dev)).cpu().numpy()[0]
            ref = z[f"f{fi}_depth"]
            err = np.abs(d - ref).max() / max(1.0, float(ref.max()))
            assert err <= tol, (prec, rep, fi, err)
    with pytest.raises(Exception):
        eng(_t(np.stack([z["f0_x"], z["f1_x"]]), dev))       # a VDA engine is one stream: batch must be 1
    eng.close()


def test_vda_longer_than_window_vs_oracle(dev):
    """40 frames (> 32-frame window: the ring wraps) against the numpy oracle, fp32."""
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.vda_weights import make_vda_weights
    from oracle import d2s_oracle as O
    from oracle.vda_oracle import VideoDepthOracle
    cfg = MODELS["tiny"]
    w = make_vda_weights(cfg, 0)
    eng = ops.Engine(cfg, w, 42, 84, 1, "fp32", temporal=True)
    orc = VideoDepthOracle(cfg, w)
    for fi in range(40):
        frame = synth.structured_frame(90, 160, 200 + fi)
        x = O.normalise(O.resize_patch_aligned(np.ascontiguousarray(frame.transpose(2, 0, 1)), 84))
        d = eng(_t(x, dev)).cpu().numpy()[0]
        ref = orc.forward(x)
        err = np.abs(d - ref).max() / max(1.0, float(ref.max()))
        assert err <= 3e-4, (fi, err)
    eng.close()


@pytest.mark.parametrize("prec,tol", [("fp32", 3e-4), ("bf16", 0.036)])
def test_vda_window_wrap_vs_reference(dev, golden_dir, prec, tol):
    """40 frames of the REFERENCE's own streaming VideoDepthAnything (tests/golden/vda_tiny_long, make_golden_vda.py):
    frames 32..39 run after the 32-frame window has wrapped, so the in-place ring (oldest slot overwritten, projected
    k' | v' rows) is held to the reference's update_cache shift order (vda2_s.py:177-187), not to the restatement."""
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.vda_weights import make_vda_weights
    from oracle import d2s_oracle as O
    cfg = MODELS["tiny"]
    z = np.load(os.path.join(golden_dir, "vda_tiny_long.npz"))
    meta = json.load(open(os.path.join(golden_dir, "vda_tiny_long.json")))
    assert len(meta["frames"]) >= 34
    eng = ops.Engine(cfg, make_vda_weights(cfg, 0), 42, 84, 1, prec, temporal=True)
    worst = 0.0
    for fi, fr in enumerate(meta["frames"]):
        frame = synth.structured_frame(fr["h"], fr["w"], fr["seed"])
        x = O.normalise(O.resize_patch_aligned(np.ascontiguousarray(frame.transpose(2, 0, 1)), meta["depth_resolution"]))
        d = eng(_t(x, dev)).cpu().numpy()[0]
        ref = z[f"f{fi}_depth"]
        err = np.abs(d - ref).max() / max(1.0, float(ref.max()))
        worst = max(worst, err)
        assert err <= tol, (prec, fi, err)
    print(f"[vda window wrap, {prec}] worst frame error {worst:.2e} of the range over {len(meta['frames'])} frames")
    eng.close()


def test_vda_vits_stream(dev, golden_dir):
    """ViT-S VDA at 196x336 (BASELINE config 4 shape), 3 frames, reference goldens."""
    path = os.path.join(golden_dir, "vda_vits.npz")
    if not os.path.exists(path):
        pytest.skip("vda_vits golden not generated")
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.vda_weights import make_vda_weights
    cfg = MODELS["vits"]
    z = np.load(path)
    meta = json.load(open(os.path.join(golden_dir, "vda_vits.json")))
    eng = ops.Engine(cfg, make_vda_weights(cfg, 0), 196, 336, 1, "fp32", temporal=True)
    for fi, fr in enumerate(meta["frames"]):
        x = ops.preprocess(_t(synth.structured_frame(fr["h"], fr["w"], fr["seed"]), dev), meta["depth_resolution"])
        d = eng(x).cpu().numpy()[0]
        ref = z[f"f{fi}_depth"]
        err = np.abs(d - ref).max() / max(1.0, float(ref.max()))
        assert err <= 3e-4, (fi, err)
    eng.close()


@pytest.mark.parametrize("prec,tol", [("fp32", 3e-4), ("bf16", 0.036)])
def test_vda_window_wrap_at_real_dimensions(dev, golden_dir, prec, tol):
    """The 32-frame window wrapping AT SIZE: 40 frames of the REFERENCE's streaming VideoDepthAnything, ViT-S at 196 x 336 on 1080p
    frames (tests/golden/vda_vits_long: every 7th depth row of every frame; caches of 1344 / 336 / 84 sites).  Frames 32..39 see a
    window whose oldest entries were evicted by the reference's update_cache shift (vda2_s.py:177-187); the HIP engine's in-place
    ring of projected rows must produce the same maps."""
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.vda_weights import make_vda_weights
    cfg = MODELS["vits"]
    z = np.load(os.path.join(golden_dir, "vda_vits_long.npz"))
    meta = json.load(open(os.path.join(golden_dir, "vda_vits_long.json")))
    rs = meta["row_stride"]
    assert len(meta["frames"]) == 40
    eng = ops.Engine(cfg, make_vda_weights(cfg, 0), 196, 336, 1, prec, temporal=True)
    worst, worst_wrapped = 0.0, 0.0
    for fi, fr in enumerate(meta["frames"]):
        x = ops.preprocess(_t(synth.structured_frame(fr["h"], fr["w"], fr["seed"]), dev), meta["depth_resolution"])
        d = eng(x).cpu().numpy()[0][::rs]
        ref = z[f"f{fi}_depth"]
        err = float(np.abs(d - ref).max() / max(1.0, float(fr["range"][1])))
        worst = max(worst, err)
        if fi >= 32:
            worst_wrapped = max(worst_wrapped, err)
        assert err <= tol, (prec, fi, err)
    print(f"[vda ViT-S 196x336 window wrap, {prec}] worst frame error {worst:.2e} of the range over 40 frames ({worst_wrapped:.2e} over frames 32-39)")
    eng.close()


@pytest.mark.parametrize("prec,tol", [("fp32", 3e-4), ("bf16x3", 3e-4), ("bf16", None)])
def test_vda_vitb_stream_at_the_quoted_size(dev, golden_dir, prec, tol):
    """ViT-B VDA at 294 x 518 on 1080p frames -- the size BASELINE config 4's stream throughput is quoted on -- 3 frames of the
    REFERENCE's own streaming model (tests/golden/vda_vitb, make_golden_vda.py): fp32 and split-precision engines within 3e-4
    of the range, the bf16 engine graded in the bf16 class (measured value printed, bound = 1.5 x it)."""
    path = os.path.join(golden_dir, "vda_vitb.npz")
    assert os.path.exists(path), "tests/golden/vda_vitb.npz is part of the repo (python tests/golden/make_golden_vda.py vda_vitb)"
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.vda_weights import make_vda_weights
    cfg = MODELS["vitb"]
    z = np.load(path)
    meta = json.load(open(os.path.join(golden_dir, "vda_vitb.json")))
    eng = ops.Engine(cfg, make_vda_weights(cfg, 0), 294, 518, 1, prec, temporal=True)
    worst = 0.0
    for fi, fr in enumerate(meta["frames"]):
        x = ops.preprocess(_t(synth.structured_frame(fr["h"], fr["w"], fr["seed"]), dev), meta["depth_resolution"])
        d = eng(x).cpu().numpy()[0]
        ref = z[f"f{fi}_depth"]
        worst = max(worst, float(np.abs(d - ref).max() / max(1.0, float(ref.max()))))
    print(f"[vda ViT-B 294x518, {prec}] worst frame error {worst:.2e} of the range over {len(meta['frames'])} frames")
    assert worst <= (tol if tol is not None else VDA_VITB_BF16_BOUND), (prec, worst)
    eng.close()


VDA_VITB_BF16_BOUND = 0.0224    # 1.5 x the measured 0.0149 (MI355X, round 3)
