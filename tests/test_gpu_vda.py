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
    of the range.  The bf16 engine's bound is REFERENCE-DERIVED (round 5): tests/golden/vda_vitb_bf16 is the reference on the same
    frames with its autocast on (forward(fp32=False), the `FP16: true` setting, vda2_s.py:193 -- bf16 on the CPU); on every
    frame the HIP bf16 engine must be no further from the reference's fp32 result than the reference's own reduced-precision path is
    (max and mean), as tests/test_gpu_parity.py::test_bf16_engine_within_reference_bf16_class does for DA-v2."""
    path = os.path.join(golden_dir, "vda_vitb.npz")
    assert os.path.exists(path), "tests/golden/vda_vitb.npz is part of the repo (python tests/golden/make_golden_vda.py vda_vitb)"
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.vda_weights import make_vda_weights
    cfg = MODELS["vitb"]
    z = np.load(path)
    zb = np.load(os.path.join(golden_dir, "vda_vitb_bf16.npz"))
    meta = json.load(open(os.path.join(golden_dir, "vda_vitb.json")))
    meta_b = json.load(open(os.path.join(golden_dir, "vda_vitb_bf16.json")))
    assert [f["seed"] for f in meta["frames"]] == [f["seed"] for f in meta_b["frames"]] and meta_b["autocast"].startswith("on")
    eng = ops.Engine(cfg, make_vda_weights(cfg, 0), 294, 518, 1, prec, temporal=True)
    worst = 0.0
    pooled = []
    for fi, fr in enumerate(meta["frames"]):
        x = ops.preprocess(_t(synth.structured_frame(fr["h"], fr["w"], fr["seed"]), dev), meta["depth_resolution"])
        d = eng(x).cpu().numpy()[0]
        ref = z[f"f{fi}_depth"]
        scale = max(1.0, float(ref.max()))
        err = np.abs(d - ref) / scale
        worst = max(worst, float(err.max()))
        if tol is None:
            gap = np.abs(zb[f"f{fi}_depth"].astype(np.float32) - ref) / scale
            print(f"[vda ViT-B 294x518, bf16] frame {fi}: HIP max {err.max():.4f} mean {err.mean():.5f} | the reference's own autocast path "
                  f"max {gap.max():.4f} mean {gap.mean():.5f} (of the range)")
            pooled.append((float(err.max()), float(err.mean()), float(gap.max()), float(gap.mean())))
    print(f"[vda ViT-B 294x518, {prec}] worst frame error {worst:.2e} of the range over {len(meta['frames'])} frames")
    if tol is not None:
        assert worst <= tol, (prec, worst)
    else:       # over the three frames: worst max and average mean, HIP against the reference's own reduced-precision path
        a = np.array(pooled)
        # measured (MI355X, end of round 5, LayerNorm / GEGLU folded into the temporal linears): per frame HIP max 0.0133 / 0.0139 / 0.0120,
        # mean 0.00225 / 0.00260 / 0.00231 against the reference's 0.0154 / 0.0153 / 0.0146 and 0.00242 / 0.00269 / 0.00241 -- inside
        # the reference's own envelope on every frame, so the gate is the strict one, per frame (round 4's engine sat 0.6 % above on the mean)
        for fi, (hmax, hmean, rmax, rmean) in enumerate(pooled):
            assert hmax <= rmax and hmean <= rmean, (fi, hmax, hmean, rmax, rmean)
    eng.close()


def test_vda_stream_through_pipeline_at_1080p(dev, golden_dir):
    """What `bench.py --vda` times: a temporal engine driven through d2s_pipeline (pre-process -> streaming forward -> post-process ->
    warp) on 1080p frames, EMA off, 40 frames (the window wraps).  Engine A runs the bare forward per frame and is held to the
    REFERENCE's rows (tests/golden/vda_vits_long); engine B runs the same frames through d2s_pipeline: its full-resolution depth must
    equal oracle post-process + up-sample of A's raw map (<= 2e-5: same kernels, the stream state is the only thing that could
    differ), and its packed Full-SBS frame the oracle's make_sbs of that depth to <= 1 LSB (checked on frames before, at and after
    the wrap)."""
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams
    from desktop2stereo_amd.vda_weights import make_vda_weights
    from oracle import d2s_oracle as O
    cfg = MODELS["vits"]
    z = np.load(os.path.join(golden_dir, "vda_vits_long.npz"))
    meta = json.load(open(os.path.join(golden_dir, "vda_vits_long.json")))
    rs = meta["row_stride"]
    res = meta["depth_resolution"]
    p = PipelineParams(depth_resolution=res)
    sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, "Full-SBS", p.fill_16_9)
    wts = make_vda_weights(cfg, 0)
    eng_a = ops.Engine(cfg, wts, 196, 336, 1, "fp32", temporal=True)
    eng_b = ops.Engine(cfg, wts, 196, 336, 1, "fp32", temporal=True)
    worst_d, worst_lsb, n_over = 0.0, 0, 0
    for fi, fr in enumerate(meta["frames"]):
        frame = synth.structured_frame(fr["h"], fr["w"], fr["seed"])
        ft = _t(frame, dev)
        raw = eng_a(ops.preprocess(ft, res)).cpu().numpy()[0]
        err = float(np.abs(raw[::rs] - z[f"f{fi}_depth"]).max() / max(1.0, float(fr["range"][1])))
        assert err <= 3e-4, ("forward vs the reference", fi, err)
        out, dfull = eng_b.pipeline(ft[None], p, sp, use_ema=False, want_depth=True)
        want_post = O.post_process_depth(raw, p.foreground_scale, p.aa_strength)
        want_full = O.upsample_depth(want_post, fr["h"], fr["w"])
        dd = float(np.abs(dfull.cpu().numpy()[0] - want_full).max())
        worst_d = max(worst_d, dd)
        assert dd <= 2e-5, ("pipeline depth vs forward + oracle post-process", fi, dd)
        if fi in (0, 1, 31, 32, 39):
            # (the warp alone: the pipeline's own full-resolution depth on both sides, as tests/test_gpu_configs.py does)
            want = O.to_u8(O.make_sbs_core(frame.transpose(2, 0, 1).astype(np.float32), dfull.cpu().numpy()[0], p.ipd, p.depth_strength,
                                           "Full-SBS", p.fill_16_9, p.convergence).transpose(1, 2, 0))
            diff = np.abs(out.cpu().numpy()[0].astype(np.int32) - want.astype(np.int32))
            worst_lsb = max(worst_lsb, int(diff.max()))
            n_over += int((diff > 1).sum())
            assert diff.max() <= 1, ("pipeline warp vs the oracle", fi, int(diff.max()))
    print(f"[vda stream through d2s_pipeline, 1080p, 40 frames] depth vs forward + oracle post-process max {worst_d:.2e}; "
          f"Full-SBS vs the oracle warp max {worst_lsb} LSB")
    eng_a.close()
    eng_b.close()


def test_vda_fused_modules_agree_with_the_separate_launches(dev, monkeypatch):
    """the round-5 fusions of the temporal modules (LayerNorms folded into the linears either side, GEGLU in ff1's epilogue,
    the ring store inside the attention kernel; D2S_VDA_FUSE, read when the engine is finalised) and the register-resident GroupNorm
    (D2S_GN_OLD) are pinned against the launches they replace, per frame over a 40-frame stream (> the 32-frame window: the in-kernel
    ring store overwrites live slots).  fp32 engines: 2e-5 of the range.  bf16 engines: the LayerNorm fold and GEGLU-in-epilogue change
    where values are rounded to bf16, and with seeded random weights ANY rounding-level change re-draws the engine's precision noise
    (a 1e-7 input perturbation moves the bf16 depth by 0.0024 mean / 0.020 max) -- so the bf16 comparison is held
    to that floor (max 3e-2, mean 5e-3 of the range per frame; a packing / column-sum bug is O(1)), and the figures are printed."""
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.vda_weights import make_vda_weights
    from oracle import d2s_oracle as O
    keys = ("D2S_VDA_FUSE", "D2S_GN_OLD")

    def stream(cfg, w, h, wd, prec, env, frames):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ops.reload_env()
        eng = ops.Engine(cfg, w, h, wd, 1, prec, temporal=True)
        outs = [eng(_t(x, dev)).cpu().numpy()[0] for x in frames]
        eng.close()
        return outs

    try:
        for name, (h, wd, res), nfr in (("tiny", (42, 84, 84), 40), ("vits", (196, 336, 336), 36)):
            cfg = MODELS[name]
            w = make_vda_weights(cfg, 0)
            src_h, src_w = (90, 160) if name == "tiny" else (360, 640)
            frames = [O.normalise(O.resize_patch_aligned(np.ascontiguousarray(synth.structured_frame(src_h, src_w, 300 + i).transpose(2, 0, 1)), res))
                      for i in range(nfr)]
            assert frames[0].shape[-2:] == (h, wd)
            for prec, tol_max, tol_mean in (("fp32", 2e-5, 2e-5), ("bf16", 3e-2, 5e-3)):
                fused = stream(cfg, w, h, wd, prec, {}, frames)
                plain = stream(cfg, w, h, wd, prec, {"D2S_VDA_FUSE": "0"}, frames)
                gn_old = stream(cfg, w, h, wd, prec, {"D2S_GN_OLD": "1"}, frames)
                worst = {"D2S_VDA_FUSE": [0.0, 0.0], "D2S_GN_OLD": [0.0, 0.0]}
                for fi in range(nfr):
                    rng = max(1.0, float(plain[fi].max()))
                    for key, other in (("D2S_VDA_FUSE", plain), ("D2S_GN_OLD", gn_old)):
                        d = np.abs(fused[fi] - other[fi]) / rng
                        worst[key] = [max(worst[key][0], float(d.max())), max(worst[key][1], float(d.mean()))]
                        assert d.max() <= tol_max and d.mean() <= tol_mean, (name, prec, fi, key, float(d.max()), float(d.mean()))
                print(f"[vda A/B {name} {prec}, {nfr} frames] fused vs separate launches: max {worst['D2S_VDA_FUSE'][0]:.2e} mean {worst['D2S_VDA_FUSE'][1]:.2e}"
                      f" | GroupNorm new vs old: max {worst['D2S_GN_OLD'][0]:.2e} mean {worst['D2S_GN_OLD'][1]:.2e}")
    finally:
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        ops.reload_env()
