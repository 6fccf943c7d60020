"""GPU: the e4m3 path of BASELINE config 3 (D2S_PREC_FP8).  The reference has no fp8 mode (FP16 only), so SURVEY.md
section 8d makes fp8 a REPORTED deviation, not a parity gate: the GEMM itself is held to an exact emulation (operands
rounded to OCP e4m3fn, float accumulate), the encoder conversions to torch's float8_e4m3fn cast, and the whole engine
is graded against the fp32 reference goldens with a loose ceiling."""
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


def test_e4m3_gemm_vs_emulation(dev):
    from desktop2stereo_amd import ops
    torch.manual_seed(0)
    for (M, N, K) in [(778, 768, 768), (300, 2304, 3072), (333, 128, 256)]:
        A = torch.randn(M, K) * 3
        W = torch.randn(N, K)
        b = torch.randn(N)
        A[0, :8] = torch.tensor([500., -500., 448., 1e-3, 2e-3, 0.0175, -0.0009, 464.])   # saturation / subnormals / ties
        Aq = A.to(torch.float8_e4m3fn).float()
        Aq = torch.where(A.abs() >= 448, torch.sign(A) * 448, Aq)                            # the library saturates (torch: NaN)
        Wq = W.to(torch.float8_e4m3fn).float()
        ref = (Aq.double() @ Wq.double().T + b.double()).float()
        for tile in (0, 64, 3264, 964, 91288):
            out = ops.gemm_probe(A.to(dev), W.to(dev), b.to(dev), "fp8", tile).cpu()
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 1e-4, (M, N, K, tile, err)          # float32 accumulation order only


def test_fp8_engine_needs_calibration_and_tracks_fp32(dev, golden_dir):
    from desktop2stereo_amd import _lib, ops, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
    from desktop2stereo_amd.weights import make_weights
    z = np.load(os.path.join(golden_dir, "vits_r518.npz"))
    meta = json.load(open(os.path.join(golden_dir, "vits_r518.json")))
    fr = meta["frames"][0]
    cfg = MODELS["vits"]
    h, w, _ = engine_shape(fr["h"], fr["w"], 518)
    x = ops.preprocess(torch.from_numpy(synth.structured_frame(fr["h"], fr["w"], fr["seed"])).to(dev), 518)
    eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, 2, "fp8")
    with pytest.raises(_lib.D2SError):
        eng(x)                                                # not calibrated yet
    calib = torch.cat([x, ops.preprocess(torch.from_numpy(synth.structured_frame(fr["h"], fr["w"], 77)).to(dev), 518)])
    eng.calibrate(calib)
    raw = eng(x)
    p = PipelineParams()
    post = ops.post_process_depth(raw, p).cpu().numpy()[0]
    d = np.abs(post - z["f0_post_depth"])
    print(f"[vits fp8] post-depth vs fp32 reference: max {d.max():.4f} mean {d.mean():.5f}")
    # fp8 deviation is reported, not gated; the bound only catches breakage.  (0.0196 with separate LN kernels,
    # 0.0207 with LayerNorm folded into the e4m3 linears -- the raw residual is what gets quantised then)
    assert d.mean() <= 0.025 and d.max() <= 0.25, (d.mean(), d.max())
    # batch of two == two single calls (static scales: no cross-frame coupling)
    both = eng(calib).cpu().numpy()
    one = eng(calib[1:2]).cpu().numpy()[0]
    assert np.abs(both[1] - one).max() <= 1e-4 * float(np.abs(one).max())
    # the bf16 engine is the yardstick the fp8 deviation is reported against
    engb = ops.Engine(cfg, make_weights(cfg, 0), h, w, 1, "bf16")
    postb = ops.post_process_depth(engb(x), p).cpu().numpy()[0]
    db = np.abs(postb - z["f0_post_depth"])
    print(f"[vits bf16] post-depth vs fp32 reference: max {db.max():.4f} mean {db.mean():.5f}")
    eng.close(); engb.close()


def test_fp8_through_the_call_surface(dev):
    """configure(precision="fp8"): the engine calibrates itself on the first frame it sees, then predict_depth /
    pipeline behave as usual (ViT-L dims are not needed for the plumbing: KAT-tiny model)."""
    from desktop2stereo_amd import depth as D, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams
    from desktop2stereo_amd.weights import make_weights
    from oracle import d2s_oracle as O
    try:
        D.configure("tiny", params=PipelineParams(depth_resolution=140), precision="fp8", max_batch=2)
        f = synth.structured_frame(270, 480, 5)
        d = D.predict_depth(f, use_temporal_smooth=False).cpu().numpy()
        cfg = MODELS["tiny"]
        ref = O.PipelineOracle(cfg, make_weights(cfg, 0), 140).predict_depth(f)
        err = np.abs(d - ref)
        print(f"[tiny fp8 predict_depth] max {err.max():.4f} mean {err.mean():.5f}")
        assert err.mean() <= 0.05                              # reported deviation class (2-head, 19-token KAT model)
        D.configure("tiny", params=PipelineParams(depth_resolution=140), precision="fp8", max_batch=2)
        out = D.pipeline(np.stack([f, synth.structured_frame(270, 480, 6)]), display_mode="Half-TAB")
        assert tuple(out.shape) == (2, 270, 480, 3) and out.dtype == torch.uint8
        # explicit calibration on representative frames instead of "whatever arrived first": a black first frame would have
        # left ranges that real frames saturate; after calibrate() on two structured frames the result is the same class
        D.configure("tiny", params=PipelineParams(depth_resolution=140), precision="fp8", max_batch=2)
        D.calibrate(np.stack([f, synth.structured_frame(270, 480, 6)]))
        d2 = D.predict_depth(f, use_temporal_smooth=False).cpu().numpy()
        assert np.abs(d2 - ref).mean() <= 0.05
    finally:
        D.configure("tiny", params=PipelineParams(depth_resolution=140), precision="fp32", max_batch=4)


def test_fp8_batched_tiles_agree_with_small_tiles(dev, golden_dir, monkeypatch):
    """The batched e4m3 regime -- 256 x 256 ping-pong tiles on v_mfma_scale_f32_32x32x64_f8f6f4, FC1 writing e4m3 from the ping-pong
    epilogue, the proj / FC2 column vectors through LDS (round 5) -- against the SAME engine on the SAME 12 frames with the ping-pong
    kernel switched off (D2S_GEMM_PP=0: 128 x 128 tiles on v_mfma_f32_16x16x32_fp8_fp8): same static scales, same quantisation points,
    only the fp32 summation order differs.  Both schemes; 12 frames of ViT-B = 9 336 rows = 111 / 333 / 444 tiles per launch.
    What "agree" can mean here was measured first: this random-weight network turns ANY rounding-level change into a fresh sample of
    its precision's noise -- perturbing the input by 1e-7 relative moves the e4m3 engine's post-processed depth by 0.0085 mean / 0.06
    max (bf16 engine: 0.0024 / 0.02), the same for 1e-4.  So the yardstick is taken in the test: the small-tile engine against itself
    on an input perturbed by 1e-6; the two kernels must not differ by more than that floor (x 1.25 / 1.5), and the batched result must
    be as close to the reference's fp32 depth as the small-tile one (x 1.15).  A layout or scale error in the new epilogues is not
    rounding-sized: it fails all three.  (Batch <= 3 is a different scheme -- LayerNorm folded, the raw residual quantised.)"""
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
    from desktop2stereo_amd.weights import make_weights
    z = np.load(os.path.join(golden_dir, "vitb_r518.npz"))
    fr = json.load(open(os.path.join(golden_dir, "vitb_r518.json")))["frames"][0]
    cfg = MODELS["vitb"]
    h, w, _ = engine_shape(fr["h"], fr["w"], 518)
    B = 12
    frames = [synth.structured_frame(fr["h"], fr["w"], fr["seed"])] + [synth.structured_frame(fr["h"], fr["w"], 300 + i) for i in range(B - 1)]
    x = torch.cat([ops.preprocess(torch.from_numpy(f).to(dev), 518) for f in frames])
    xn = x * (1 + 1e-6 * torch.randn(x.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1)))
    p = PipelineParams()
    try:
        for prec in ("fp8", "fp8_mlp"):
            eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, B, prec)
            eng.calibrate(x[:2])
            monkeypatch.delenv("D2S_GEMM_PP", raising=False); ops.reload_env()
            big = ops.post_process_depth(eng(x), p).cpu().numpy()
            monkeypatch.setenv("D2S_GEMM_PP", "0"); ops.reload_env()
            small = ops.post_process_depth(eng(x), p).cpu().numpy()
            floor = np.abs(ops.post_process_depth(eng(xn), p).cpu().numpy() - small)
            d = np.abs(big - small)
            dev_small, dev_big = np.abs(small[0] - z["f0_post_depth"]), np.abs(big[0] - z["f0_post_depth"])
            print(f"[vitb {prec}] 256x256 K64 tiles vs small tiles: mean {d.mean():.5f} max {d.max():.4f}; rounding floor (input x (1 + 1e-6 n)): mean {floor.mean():.5f} "
                  f"max {floor.max():.4f}; vs fp32 reference: small {dev_small.mean():.5f} / {dev_small.max():.4f}, batched {dev_big.mean():.5f} / {dev_big.max():.4f}")
            assert d.mean() <= 1.25 * floor.mean() and d.max() <= 1.5 * floor.max(), (prec, d.mean(), d.max(), floor.mean(), floor.max())
            assert dev_big.mean() <= 1.15 * dev_small.mean() + 1e-4, (prec, dev_big.mean(), dev_small.mean())
            eng.close()
    finally:
        monkeypatch.delenv("D2S_GEMM_PP", raising=False); ops.reload_env()
