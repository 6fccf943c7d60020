"""CPU: the numpy oracle (oracle/d2s_oracle.py) against the golden vectors captured from the
reference itself (tests/golden/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest

from desktop2stereo_amd import synth
from desktop2stereo_amd.config import MODELS
from desktop2stereo_amd.weights import make_weights
from oracle import d2s_oracle as O


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    with open(os.path.join(golden_dir, name + ".json")) as f:
        meta = json.load(f)
    return z, meta


@pytest.fixture(scope="module")
def tiny():
    cfg = MODELS["tiny"]
    return cfg, make_weights(cfg, 0)


def test_engine_shape_table():
    # reference depth.py:676-706 integer logic; 16:9 frames all land on 294x518 at T=518
    assert O.engine_shape(1080, 1920, 518) == (294, 518, 1)
    assert O.engine_shape(2160, 3840, 518) == (294, 518, 3)
    assert O.engine_shape(1440, 2560, 518) == (294, 518, 2)
    assert O.engine_shape(720, 1280, 518) == (294, 518, 1)
    assert O.engine_shape(1080, 1920, 336) == (196, 336, 2)
    assert O.engine_shape(90, 160, 84) == (42, 84, 1)


def test_tiny_all_taps(golden_dir, tiny):
    cfg, w = tiny
    z, meta = _load(golden_dir, "tiny_r84")
    orc = O.PipelineOracle(cfg, w, 84)
    for fi in range(len(meta["frames"])):
        p = f"f{fi}_"
        img = z[p + "img"]
        fr = meta["frames"][fi]
        gen = synth.structured_frame if fr["kind"] == "S2" else synth.noise_frame
        assert np.array_equal(img, gen(fr["h"], fr["w"], fr["seed"]))      # generators are stable
        x = orc.model_input(img)
        np.testing.assert_allclose(x, z[p + "model_input"], atol=2e-5)
        taps = {}
        raw = orc.model.forward(z[p + "model_input"], taps)
        np.testing.assert_allclose(taps["embeddings"], z[p + "embeddings"], atol=2e-5)
        for li in range(1, cfg.layers + 1):
            np.testing.assert_allclose(taps[f"layer{li}"], z[p + f"layer{li}"], atol=5e-5)
        scale = float(z[p + "raw_depth"].max())
        assert np.abs(raw - z[p + "raw_depth"]).max() <= 2e-5 * scale
        rawg = z[p + "raw_depth"]
        nrm = O.normalize_depth(rawg)
        np.testing.assert_allclose(nrm, z[p + "norm"], atol=1e-6)
        gam = O.apply_gamma(z[p + "norm"])
        np.testing.assert_allclose(gam, z[p + "gamma"], atol=1e-6)
        fg = O.apply_foreground_scale(z[p + "gamma"], 0.05)
        np.testing.assert_allclose(fg, z[p + "fg"], atol=1e-6)
        post = O.anti_alias(z[p + "fg"], 4.0)
        np.testing.assert_allclose(post, z[p + "post_depth"], atol=1e-6)
        np.testing.assert_allclose(O.post_process_depth(rawg), z[p + "post_depth"], atol=2e-6)


def test_tiny_ema_chain(golden_dir, tiny):
    """predict_depth with use_temporal_smooth=True over 3 frames (reference depth.py:1865-1887)."""
    cfg, w = tiny
    z, meta = _load(golden_dir, "tiny_r84")
    orc = O.PipelineOracle(cfg, w, 84)
    for fi in range(len(meta["frames"])):
        d = orc.predict_depth(z[f"f{fi}_img"], use_temporal_smooth=True)
        np.testing.assert_allclose(orc.stab.prev, z[f"f{fi}_ema_state"], atol=2e-5)
        np.testing.assert_allclose(d, z[f"f{fi}_depth_ema_full"], atol=2e-5)


@pytest.mark.parametrize("name,model,res", [("tiny_r518", "tiny", 518)])
def test_full_size_model(golden_dir, name, model, res):
    cfg = MODELS[model]
    z, meta = _load(golden_dir, name)
    fr = meta["frames"][0]
    orc = O.PipelineOracle(cfg, make_weights(cfg, 0), res)
    taps = {}
    orc.predict_depth(synth.structured_frame(fr["h"], fr["w"], fr["seed"]), taps=taps)
    scale = float(z["f0_raw_depth"].max())
    assert np.abs(taps["raw_depth"] - z["f0_raw_depth"]).max() <= 2e-5 * scale
    assert np.abs(taps["post_depth"] - z["f0_post_depth"]).max() <= 1e-4


def test_cuda_branch_preprocess(golden_dir):
    """_resize_patch_aligned_t's IS_CUDA branch (reference depth.py:698-699: bicubic + antialias from the full frame, the
    branch the reference takes on a ROCm device) as the reference executed it with IS_CUDA forced on: 1080p, 4K (no ::3
    decimation on this branch), 1440p, 720p and an odd size; then the ViT-S depth of the 1080p frame."""
    z, meta = _load(golden_dir, "vits_r518_cuda")
    assert meta["cuda_branch"]
    for fi, fr in enumerate(meta["frames"]):
        gen = synth.structured_frame if fr["kind"] == "S2" else synth.noise_frame
        img = gen(fr["h"], fr["w"], fr["seed"])
        got = O.resize_patch_aligned(np.ascontiguousarray(img.transpose(2, 0, 1)), meta["depth_resolution"], cuda_branch=True)
        ref = z[f"f{fi}_resized_rows"]
        assert got[:, ::14].shape == ref.shape, (fi, got.shape)
        assert np.abs(got[:, ::14] - ref).max() <= 3e-4, (fi, np.abs(got[:, ::14] - ref).max())    # of 255: summation order
    cfg = MODELS["vits"]
    fr = meta["frames"][0]
    orc = O.PipelineOracle(cfg, make_weights(cfg, 0), 518, cuda_branch=True)
    taps = {}
    orc.predict_depth(synth.structured_frame(fr["h"], fr["w"], fr["seed"]), taps=taps)
    scale = float(z["f0_raw_depth"].max())
    assert np.abs(taps["raw_depth"] - z["f0_raw_depth"]).max() <= 3e-5 * scale
    assert np.abs(taps["post_depth"] - z["f0_post_depth"]).max() <= 1e-4


def test_metric_model_and_normalize(golden_dir, tiny):
    """Depth-Anything-V2-Metric-* ids: sigmoid * max_depth head (HF) and normalize()'s is_metric() branch
    (reference depth.py:844-847) incl. maps with invalid pixels and the <= 10-valid-values rule."""
    cfg, w = tiny
    z, meta = _load(golden_dir, "tiny_r84_metric")
    assert meta["metric"] == "Indoor" and meta["max_depth"] == 20.0
    orc = O.PipelineOracle(cfg, w, 84, metric=True, max_depth=meta["max_depth"])
    for fi in range(len(meta["frames"])):
        p = f"f{fi}_"
        raw = orc.model.forward(z[p + "model_input"])
        # logits carry the usual ~2e-5 * |logit| float32 noise; the sigmoid * 20 has slope up to 5
        assert np.abs(raw - z[p + "raw_depth"]).max() <= 1.5e-3
        np.testing.assert_allclose(O.normalize_depth(z[p + "raw_depth"], metric=True), z[p + "norm"], atol=2e-6)
        np.testing.assert_allclose(O.post_process_depth(z[p + "raw_depth"], metric=True), z[p + "post_depth"], atol=3e-6)
        d = orc.predict_depth(z[p + "img"], use_temporal_smooth=True)
        np.testing.assert_allclose(d, z[p + "depth_ema_full"], atol=3e-4)       # 1/d amplifies the 2e-5 model noise
    np.testing.assert_allclose(O.normalize_depth(z["normcase_in"], metric=True), z["normcase_norm"], atol=1e-6)
    np.testing.assert_allclose(O.post_process_depth(z["normcase_in"], metric=True), z["normcase_post"], atol=2e-6)
    np.testing.assert_allclose(O.normalize_depth(z["fewcase_in"], metric=True), z["fewcase_norm"], atol=1e-6)


def test_fixed_square_branch(golden_dir, tiny):
    """predict_depth's fixed-square input branch (get_patch_size() is None: CAPTURE_MODE == "Window", reference depth.py:531-538,
    1937-1946) as the reference ran it: the model input it fed its model (captured at the call), every tap of the KAT model at
    84 x 84 incl. a frame that already is 84 x 84, and ViT-S at 518 x 518 from a 1080p frame."""
    cfg, w = tiny
    z, meta = _load(golden_dir, "tiny_r84_square")
    assert meta["square"]
    orc = O.PipelineOracle(cfg, w, 84, square=True)
    for fi, fr in enumerate(meta["frames"]):
        p = f"f{fi}_"
        x = orc.model_input(z[p + "img"])
        assert x.shape == (3, 84, 84)
        np.testing.assert_allclose(x, z[p + "model_input"], atol=2e-5)
        taps = {}
        d = orc.predict_depth(z[p + "img"], use_temporal_smooth=True, taps=taps)      # the EMA chain runs across both frames (84 x 84 state)
        for li in range(1, cfg.layers + 1):
            np.testing.assert_allclose(taps[f"layer{li}"], z[p + f"layer{li}"], atol=5e-5)
        assert np.abs(taps["raw_depth"] - z[p + "raw_depth"]).max() <= 2e-5 * float(z[p + "raw_depth"].max())
        np.testing.assert_allclose(taps["post_depth"], z[p + "post_depth"], atol=1e-4)
        np.testing.assert_allclose(d, z[p + "depth_ema_full"], atol=1e-4)
    z, meta = _load(golden_dir, "vits_r518_square")
    cfg = MODELS["vits"]
    fr = meta["frames"][0]
    orc = O.PipelineOracle(cfg, make_weights(cfg, 0), 518, square=True)
    taps = {}
    orc.predict_depth(synth.structured_frame(fr["h"], fr["w"], fr["seed"]), taps=taps)
    np.testing.assert_allclose(taps["model_input"][:, ::37], z["f0_model_input_rows"], atol=2e-5)
    assert np.abs(taps["raw_depth"] - z["f0_raw_depth"]).max() <= 3e-5 * float(z["f0_raw_depth"].max())
    assert np.abs(taps["post_depth"] - z["f0_post_depth"]).max() <= 1e-4


def test_process_area_restatement():
    """process()'s cv2 branch (reference depth.py:603-629; cv2.resize INTER_AREA).  cv2 is not installed here, so this row is
    PARITY UNPINNED against OpenCV itself: the restatement of its published algorithm is held to an exact float64 area
    integral (what INTER_AREA approximates in float32) within one level, on integer (2x2, 3x3) and fractional factors."""
    rng = np.random.default_rng(3)

    def exact(img, dw, dh):
        H, W, _ = img.shape

        def mat(s, d):
            M, sc = np.zeros((d, s)), s / d
            for i in range(d):
                a, b = i * sc, (i + 1) * sc
                for j in range(int(np.floor(a)), min(s, int(np.ceil(b)))):
                    M[i, j] = max(0.0, min(b, j + 1) - max(a, j)) / sc
            return M
        return np.einsum("ij,jkc,lk->ilc", mat(H, dh), img.astype(np.float64), mat(W, dw))
    for (H0, W0, ch, t) in [(90, 160, 3, 45), (90, 160, 4, 30), (108, 192, 3, 72), (101, 75, 4, 33), (96, 128, 3, 67), (60, 80, 3, 60)]:
        img = rng.integers(0, 256, (H0, W0, ch), dtype=np.uint8)
        got = O.process_area(img, t)
        if t >= H0:
            assert np.array_equal(got, img[..., :3][..., ::-1])
            continue
        assert got.shape == (t, int(W0 * t / H0), 3) and got.dtype == np.uint8
        ex = exact(img[..., :3][..., ::-1], got.shape[1], got.shape[0])
        # right / bottom source cells beyond dsize * scale are never read by OpenCV's tables when W0 * t / H0 is not an integer
        if abs(W0 / got.shape[1] - H0 / t) < 1e-9:
            assert np.abs(got.astype(np.float64) - ex).max() <= 0.5 + 1e-3, (H0, W0, t)


def test_process_and_overlay(golden_dir):
    """A1 process(): BOTH definitions of the reference executed at generation time (ast-extracted from depth.py:540-629, never
    stored): the IS_CUDA one on BGR(A) frames (swizzle + anti-aliased bilinear down-scale) and the tensor branch of the other
    (plain bilinear, no flip); A15 overlay_fps() against the reference's function."""
    z, meta = _load(golden_dir, "ingest")
    assert "ast" in meta["process_source"]["cuda"]
    for c in meta["process_tensor"]:
        rng = np.random.default_rng(c["seed"])
        shape = (c["channels"], c["H0"], c["W0"]) if c["layout"] == "chw" else (c["H0"], c["W0"], c["channels"])
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        if c["dtype"] == "f32":
            img = img.astype(np.float32) + np.float32(0.25)
        got = O.process_tensor(img, c["target"])
        assert list(got.shape) == c["out_shape"] and str(got.dtype) == c["out_dtype"], c
        err = np.abs(got[:, ::c["row_stride"]].astype(np.float32) - z["ptensor_" + c["name"]]).max()
        # of 255, on noise frames: ATen's CPU kernel and the restatement round the source coordinate (up to ~500, float32 ulp
        # 3e-5) in a different order, and neighbouring noise pixels differ by up to 255 levels
        assert err <= 2e-3, (c["name"], err)
    for c in meta["process"]:
        img = np.random.default_rng(c["seed"]).integers(0, 256, (c["H0"], c["W0"], c["channels"]), dtype=np.uint8)
        got = O.process_frame(img, c["target"])
        assert list(got.shape) == c["out_shape"], c
        err = np.abs(got[:, ::c["row_stride"]] - z["process_" + c["name"]]).max()
        assert err <= 2e-4, (c["name"], err)                 # of 255: float32 weight rounding of the separable filter
    for c in meta["overlay"]:
        assert not c["changed_outside_box"]
        rgb = synth.structured_frame(c["H"], c["W"], c["seed"]).transpose(2, 0, 1).astype(np.float32)
        got = O.overlay_text(rgb, f"FPS: {c['fps']:.1f}")
        bh, bw = c["box"]
        assert np.array_equal(got[:, :bh, :bw], z["overlay_" + c["name"]]), c["name"]
        assert np.array_equal(got[:, bh:], rgb[:, bh:]) and np.array_equal(got[:, :, bw:], rgb[:, :, bw:])


@pytest.mark.parametrize("fixture", ["warp", "warp_uhd"])
def test_warp_all_cases(golden_dir, fixture):
    """make_sbs with a given depth: all display modes x fill_16_9 x convergence, several aspect
    ratios (pads), 1080p rows; warp_uhd: BASELINE config 3's 3840x2160 frame (Full-TAB 4320x3840, Half-TAB, both SBS).
    Values within the reference's own float32 coordinate noise, and within 1 LSB after uint8 rounding
    (SURVEY.md section 8d parity gate)."""
    z, meta = _load(golden_dir, fixture)
    cache = {}
    for c in meta["cases"]:
        k = (c["shape"], c["kind"])
        if k not in cache:
            gen = synth.structured_frame if c["kind"] == "S2" else synth.noise_frame
            cache[k] = (gen(c["h"], c["w"], c["seed"]), synth.smooth_depth(c["h"], c["w"], c["seed"]))
        img, dep = cache[k]
        if c["shape"] not in ("hd", "uhd"):
            assert np.array_equal(img, z[f"{c['shape']}_{c['kind']}_img"])
            assert np.array_equal(dep, z[f"{c['shape']}_{c['kind']}_depth"])
        rgb = img.transpose(2, 0, 1).astype(np.float32)
        out = O.make_sbs_core(rgb, dep, ipd_uv=c["ipd_uv"], depth_ratio=c["depth_ratio"],
                              display_mode=c["mode"], fill_16_9=c["fill_16_9"],
                              convergence=c["convergence"]).transpose(1, 2, 0)
        assert list(out.shape) == c["out_shape"], c
        ref = z[c["key"]].astype(np.float32) / 256.0
        got = out[::c["row_stride"]]
        # the reference's grid_sample goes through normalised float32 coordinates: its pixel-space noise grows with W
        tol = (0.08 if c["kind"] == "S1" else 0.03) * max(1.0, c["w"] / 1920.0)
        assert np.abs(got - ref).max() <= tol, (c["key"], np.abs(got - ref).max())
        assert np.abs(O.to_u8(got).astype(int) - O.to_u8(ref).astype(int)).max() <= 1


def test_dibr_oracle_matches_reference_shader_renders(golden_dir):
    """f1, pinned (round 5): oracle/dibr_oracle.py against renders of the REFERENCE's own fragment shader (viewer.py:386-631) --
    tests/golden/dibr.npz, produced by compiling the shader text as OpenGL ES 3.0 and running it off-screen on SwiftShader in the
    build container (tests/golden/gl_harness.py, make_golden_dibr.py).  frag_color.rgb and frag_color.a separately, both eyes:
    hard depth edges, convergence, roll, feathering + rounded corners, a Half-SBS viewport, a smooth scene (small cases, every
    pixel), a Half-SBS viewport with roll, 1080p Full and Half-TAB viewports (every 45th row), a 3840x2160 frame (every 120th row).  Tolerance: GL_LINEAR filters RGB8 with 8-bit sub-texel weights
    (<= 255/512 of a level per lerp axis) where the restatement filters in float32 -> <= 1 level on the small cases (measured max
    0.48); at 1920 columns the shader's hard thresholds flip isolated pixels on a 1-ulp coordinate difference -> >= 99.9 % within
    1 level (measured 3e-4..6e-4 beyond), mean <= 0.06; alpha within 1e-3 everywhere (measured 1e-4; min alpha 0.69 at the 1080p
    screen edge and 0 in the rounded corners, so the comparison is not vacuous)."""
    from desktop2stereo_amd import synth
    from oracle import dibr_oracle as DO
    z = np.load(os.path.join(golden_dir, "dibr.npz"))
    with open(os.path.join(golden_dir, "dibr.json")) as f:
        meta = json.load(f)
    assert "SwiftShader" in meta["gl"]["renderer"] and len(meta["cases"]) >= 8
    saw_alpha = False
    for c in meta["cases"]:
        img, dep = synth.dibr_scene(c["h"], c["w"], c["seed"], c["scene"])
        for eye, sg in (("left", -1.0), ("right", 1.0)):
            o = DO.dibr_eye(img, dep, sg * c["ipd_uv"] / 2.0, 0.1 * c["depth_ratio"], c["convergence"], c["eye_h"], c["eye_w"],
                            roll=c.get("roll", 0.0), feather=c.get("feather", False), feather_width=c.get("feather_width", 0.02),
                            corner_radius=c.get("corner_radius", 0.0), rgba=True)[::c["row_stride"]]
            rgb = z[f"{c['name']}_{eye}_rgb"].astype(np.float32) / 256.0
            a = z[f"{c['name']}_{eye}_a"].astype(np.float32) / 65535.0
            d = np.abs(o[..., :3] - rgb)
            if c.get("as_shipped"):
                # the reference's as-shipped uniform state (u_resolution never assigned: pixel_size = 1 / 0, taps at non-finite
                # coordinates, undefined in GL): what SwiftShader rendered is RECORDED, the restatement (pixel_size = one texel, the
                # shader's intent and d2s_dibr_params.res_w = 0) is not held to it -- the distance is printed
                print(f"[dibr restatement vs the as-shipped render (u_resolution = 0), {c['name']} {eye}] rgb max {d.max():.1f} mean {d.mean():.3f}, "
                      f"{(d.max(-1) > 1).mean():.3f} of the pixels beyond 1 level")
                continue
            assert np.abs(o[..., 3] - a).max() <= 1e-3, (c["name"], eye)
            saw_alpha |= bool(a.min() < 0.9)
            if c["w"] <= 320:
                assert d.max() <= 1.0, (c["name"], eye, float(d.max()))
            else:
                assert (d <= 1.0).mean() >= 0.999 and d.mean() <= 0.06, (c["name"], eye, float((d > 1).mean()), float(d.mean()))
    assert saw_alpha
