"""GPU: the reference's own call surface (predict_depth / make_sbs / make_sbs_core and the
north_star aliases) on the HIP path, written like the tests the reference never had: same
arguments, same return types, checked against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def D():
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no ROCm device is visible")
    from desktop2stereo_amd import depth as D
    from desktop2stereo_amd.config import PipelineParams
    D.configure("tiny", params=PipelineParams(depth_resolution=140), precision="fp32", max_batch=4)
    return D


@pytest.fixture(scope="module")
def orc():
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.weights import make_weights
    from oracle import d2s_oracle as O
    cfg = MODELS["tiny"]
    return O.PipelineOracle(cfg, make_weights(cfg, 0), 140)


def test_predict_depth_and_make_sbs_like_the_reference(D, orc):
    from desktop2stereo_amd import synth
    from oracle import d2s_oracle as O
    frames = [synth.structured_frame(270, 480, s) for s in range(3)]
    D.depth_stabilizer.prev = None
    orc.stab.prev = None
    for f in frames:                                             # EMA on by default, like main.py:249
        d, rgb = D.predict_depth(f, return_tuple=True)
        assert d.shape == (270, 480) and d.dtype == torch.float32 and d.is_cuda
        assert rgb.shape == (3, 270, 480) and rgb.dtype == torch.uint8
        ref = orc.predict_depth(f, use_temporal_smooth=True)
        assert np.abs(d.cpu().numpy() - ref).max() <= 1e-3
        assert 0.0 <= float(d.min()) and float(d.max()) <= 1.0
        sbs = D.make_sbs(f, d, ipd_uv=0.064, depth_ratio=4.0, display_mode="Half-SBS", fill_16_9=True)
        assert isinstance(sbs, np.ndarray) and sbs.dtype == np.float32 and sbs.shape == (270, 480, 3)
        want = orc.make_sbs(f, d.cpu().numpy(), ipd_uv=0.064, depth_ratio=4.0, display_mode="Half-SBS", fill_16_9=True)
        assert np.abs(O.to_u8(sbs).astype(int) - O.to_u8(want).astype(int)).max() <= 1
    # tensor input (CHW, 0..255), smoothing off, aliases
    t = torch.from_numpy(frames[0]).permute(2, 0, 1).contiguous()
    d2 = D.predict(t, use_temporal_smooth=False)
    assert np.abs(d2.cpu().numpy() - orc.predict_depth(frames[0])).max() <= 1e-3
    full = D.to_stereo(frames[0], d2, display_mode="Full-TAB")
    assert full.shape == (540, 480, 3)
    core = D.make_sbs_core(t.float(), d2, 0.064, 2.0, "Full-SBS", False, 0.0)
    assert core.shape == (3, 270, 960) and core.dtype == torch.float32
    want = O.make_sbs_core(frames[0].transpose(2, 0, 1).astype(np.float32), d2.cpu().numpy(), 0.064, 2.0, "Full-SBS", False, 0.0)
    assert np.abs(core.cpu().numpy() - want).max() <= 0.05
    # predict_depth(dtype=): float32 / None = the kernels' map; float16 / bfloat16 = that map cast (an FP16 reference's callers get
    # DTYPE tensors, depth.py:325, 1897); anything else raises -- never silently ignored
    d32 = D.predict_depth(frames[0], use_temporal_smooth=False, dtype=torch.float32)
    assert d32.dtype == torch.float32 and torch.equal(d32, d2)
    for dt in (torch.float16, torch.bfloat16):
        dh = D.predict_depth(frames[0], use_temporal_smooth=False, dtype=dt)
        assert dh.dtype == dt and torch.equal(dh, d2.to(dt))
        assert D.make_sbs(frames[0], dh, display_mode="Half-SBS").shape == (270, 480, 3)     # make_sbs takes it back (depth.py:2209)
    with pytest.raises(TypeError):
        D.predict_depth(frames[0], dtype=torch.int32)


def test_batched_pipeline_equals_per_frame(D, orc):
    from desktop2stereo_amd import synth
    frames = np.stack([synth.structured_frame(270, 480, 10 + s) for s in range(4)])
    out, depth = D.pipeline(frames, display_mode="Full-SBS", want_depth=True)
    assert out.shape == (4, 270, 960, 3) and out.dtype == torch.uint8
    for b in range(4):
        d1 = D.predict_depth(frames[b], use_temporal_smooth=False)
        assert np.abs(depth[b].cpu().numpy() - d1.cpu().numpy()).max() <= 1e-4


def test_errors_are_loud(D):
    from desktop2stereo_amd import _lib
    with pytest.raises(ValueError):
        D.make_sbs(np.zeros((8, 8, 3), np.uint8), torch.zeros(8, 8), display_mode="Quarter-SBS")
    with pytest.raises(_lib.D2SError):
        D.pipeline(np.zeros((5, 270, 480, 3), np.uint8))          # > max_batch
    with pytest.raises(ValueError):
        D.process(np.zeros((8, 8), np.uint8), 4)                  # not HWC BGR(A)


def test_capture_to_stereo_like_main_loop(D, orc):
    """main.py's per-frame sequence: process(BGRA capture, OUTPUT_RESOLUTION) -> predict_depth(tensor) ->
    make_sbs(tensor, depth, fps=...) (reference main.py:232-262, 1336-1341; depth.py:540-566, 2216-2218)."""
    from desktop2stereo_amd import synth
    from oracle import d2s_oracle as O
    rgb_full = synth.structured_frame(540, 960, 3)
    bgra = np.concatenate([rgb_full[..., ::-1], np.full((540, 960, 1), 255, np.uint8)], -1)
    frame = D.process(bgra, 270)                                   # CHW float32 RGB, 270 x 480
    assert frame.shape == (3, 270, 480) and frame.dtype == torch.float32 and frame.is_cuda
    want_frame = O.process_frame(bgra, 270)
    assert np.abs(frame.cpu().numpy() - want_frame).max() <= 2e-4
    D.depth_stabilizer.prev = None
    d = D.predict_depth(frame, use_temporal_smooth=False)
    # the oracle's predict_depth takes uint8 HWC; feed it the same float frame through its tensor-path stages
    x = O.normalise(O.resize_patch_aligned(want_frame, 140))
    ref_d = O.upsample_depth(O.post_process_depth(orc.model.forward(x)), 270, 480)
    assert np.abs(d.cpu().numpy() - ref_d).max() <= 1e-3
    D._FPS_MASK_CACHE.update(text=None, frame=0)
    keep = frame.clone()
    sbs = D.make_sbs(frame, d, depth_ratio=4.0, display_mode="Half-SBS", fill_16_9=True, fps=58.8)
    assert torch.equal(frame, keep)                                # the caller's frame is not painted
    want = O.make_sbs_core(O.overlay_text(want_frame, "FPS: 58.8"), d.cpu().numpy(), 0.064, 4.0, "Half-SBS", True, 0.0)
    assert np.abs(O.to_u8(sbs.transpose(2, 0, 1)).astype(int) - O.to_u8(want).astype(int)).max() <= 1
    for i in range(2, 12):                                         # the text is rebuilt only every 10th call (depth.py:2069)
        D.overlay_fps(keep.clone(), 10.0 + i)
        assert D._FPS_MASK_CACHE["text"] == ("FPS: 58.8" if i < 10 else "FPS: 20.0"), i


def test_configure_metric_ids():
    """reference ids containing 'metric' switch normalize() to its 1/d branch (depth.py:666) and, for the HF
    Metric-Indoor/Outdoor checkpoints, the head to sigmoid * max_depth."""
    from desktop2stereo_amd import depth as Dm, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams
    from desktop2stereo_amd.weights import make_weights
    from oracle import d2s_oracle as O
    import dataclasses
    big = MODELS["vits"]
    cfg = dataclasses.replace(MODELS["tiny"], name="vits")          # keep the test small: tiny dims under the vits id
    try:
        MODELS["vits"] = cfg
        Dm.configure("depth-anything/Depth-Anything-V2-Metric-Indoor-Small-hf", params=PipelineParams(depth_resolution=140),
                     precision="fp32")
        f = synth.structured_frame(270, 480, 5)
        d = Dm.predict_depth(f, use_temporal_smooth=False)
        orc = O.PipelineOracle(cfg, make_weights(cfg, 0), 140, metric=True, max_depth=20.0)
        assert np.abs(d.cpu().numpy() - orc.predict_depth(f)).max() <= 1e-3
    finally:
        MODELS["vits"] = big
        Dm.configure("tiny", params=PipelineParams(depth_resolution=140), precision="fp32", max_batch=4)


def test_mixed_resolution_batch(D, orc):
    """BASELINE config 5 shape logic at KAT size: 16:9 frames of three sizes share one model-input shape."""
    from desktop2stereo_amd import synth
    from oracle import d2s_oracle as O
    sizes = [(270, 480), (180, 320), (360, 640), (270, 480)]
    assert len({O.engine_shape(h, w, 140)[:2] for h, w in sizes}) == 1
    frames = [synth.structured_frame(h, w, 20 + i) for i, (h, w) in enumerate(sizes)]
    outs = D.pipeline_mixed(frames, display_mode="Half-SBS")
    for f, o in zip(frames, outs):
        assert tuple(o.shape) == f.shape and o.dtype == torch.uint8
        d = orc.predict_depth(f)
        want = O.to_u8(orc.make_sbs(f, d, ipd_uv=0.064, depth_ratio=4.0, display_mode="Half-SBS", fill_16_9=True))
        # depth comes from the fp32 engine (<= 1e-3 of the oracle's): allow 1 LSB + rare 2-LSB flips on edges
        diff = np.abs(o.cpu().numpy().astype(int) - want.astype(int))
        assert diff.max() <= 2 and (diff > 1).mean() < 1e-3
