"""GPU: BASELINE.json's configs AS WRITTEN, at full size, through the C-ABI (d2s_pipeline / the drop-in surface).

  config 2  DepthAnything-v2 ViT-B bf16, 1920x1080, batch 1, Full-SBS      -- the exact bench.py step
  config 3  DepthAnything-v2 ViT-L fp8, 3840x2160, batch 1, Full-TAB (+ Half-TAB)
  config 5  64 frames drawn from {1280x720, 1920x1080, 2560x1440}, one model batch, per-size pre-process / warp
  + weights from a real model.safetensors written the way the reference's convert.py:14-24 does (save_pretrained)

Depth is compared with the goldens captured from the reference (fp32, tests/golden/*.npz); the warp with the oracle
fed the SAME depth (isolates A14, <= 1 LSB) and end to end (fraction of pixels off by more than 1 LSB is reported).
bf16 / fp8 depth bounds are 1.5x what was measured on MI355X (printed by the test), never looser."""
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
    from desktop2stereo_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)


def _golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    with open(os.path.join(golden_dir, name + ".json")) as f:
        return z, json.load(f)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _warp_vs_oracle(out_u8, frame, depth_full, mode, p, rows=None):
    """HIP warp output vs the oracle's make_sbs_core on the same full-resolution depth -> (max LSB, fraction > 1 LSB)."""
    from oracle import d2s_oracle as O
    want = O.to_u8(O.make_sbs_core(frame.transpose(2, 0, 1).astype(np.float32), depth_full, p.ipd, p.depth_strength, mode,
                                   p.fill_16_9, p.convergence).transpose(1, 2, 0))
    diff = np.abs(out_u8.astype(np.int16) - want.astype(np.int16))
    return int(diff.max()), float((diff > 1).mean())


def test_config2_bench_step_vitb_bf16_1080p_full_sbs(dev, golden_dir):
    """The step bench.py times: ViT-B bf16 engine (LayerNorm / tap-LN folds and the side stream ON, batch 1), 1080p uint8
    frame -> d2s_pipeline -> Full-SBS uint8 3840x1080, against the reference's fp32 depth (vitb_r518) and the oracle warp."""
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
    from desktop2stereo_amd.weights import make_weights
    from oracle import d2s_oracle as O
    z, meta = _golden(golden_dir, "vitb_r518")
    fr = meta["frames"][0]
    H, W = fr["h"], fr["w"]
    assert (H, W) == (1080, 1920)
    frame = synth.structured_frame(H, W, fr["seed"])
    cfg = MODELS["vitb"]
    h, w, _ = engine_shape(H, W, 518)
    p = PipelineParams(depth_resolution=518, display_mode="Full-SBS")
    sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, "Full-SBS", p.fill_16_9)
    ref_full = O.upsample_depth(z["f0_post_depth"], H, W)
    stats = {}
    for prec in ("fp32", "bf16x3", "bf16"):
        eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, 1, prec)
        out, depth = eng.pipeline(_t(frame[None], dev), p, sp, want_depth=True)
        out2, _ = eng.pipeline(_t(frame[None], dev), p, sp, want_depth=True)          # side stream / folds: deterministic
        assert torch.equal(out, out2)
        out, depth = out.cpu().numpy()[0], depth.cpu().numpy()[0]
        assert out.shape == (1080, 3840, 3) and out.dtype == np.uint8
        d = np.abs(depth - ref_full)
        lsb_same, frac_same = _warp_vs_oracle(out, frame, depth, "Full-SBS", p)        # warp alone: same depth both sides
        lsb_e2e, frac_e2e = _warp_vs_oracle(out, frame, ref_full, "Full-SBS", p)       # end to end vs the reference's depth
        stats[prec] = (d.max(), d.mean())
        print(f"[config 2, {prec}] depth vs reference fp32: max {d.max():.5f} mean {d.mean():.6f}; warp (same depth) max {lsb_same} LSB, "
              f"{frac_same:.2e} of bytes > 1 LSB; end to end max {lsb_e2e} LSB, {frac_e2e:.2e} of bytes > 1 LSB")
        assert lsb_same <= 1, (prec, lsb_same)
        if prec in ("fp32", "bf16x3"):               # both parity-class engines (bf16x3: the same gate on the bf16 matrix pipe)
            assert d.max() <= 1e-3, d.max()                                            # north_star's gate
            assert frac_e2e <= 1e-4, frac_e2e        # a 1e-3 depth error is < 0.03 px of shift: isolated 2-LSB flips on sharp edges
            assert lsb_e2e <= (1 if prec == "fp32" else 2), (prec, lsb_e2e)
        else:
            # reference-derived bound (round 4): the reference as shipped runs its CPU path under bf16 autocast; on this very frame it
            # sits max 0.0180 / mean 0.00355 from its own fp32 result (tests/golden/vitb_r518_bf16 vs vitb_r518, model resolution; the
            # bilinear up-sample to the frame is a convex combination, so the same bounds hold at full resolution)
            zb = np.load(os.path.join(golden_dir, "vitb_r518_bf16.npz"))
            gap = np.abs(zb["f0_post_depth"].astype(np.float32) - z["f0_post_depth"])
            print(f"[config 2] the reference's own bf16 path vs its fp32 self: max {gap.max():.5f} mean {gap.mean():.6f}")
            assert d.max() <= gap.max() and d.mean() <= gap.mean(), (d.max(), d.mean(), gap.max(), gap.mean())
        eng.close()


def test_config3_vitl_fp8_4k_tab(dev, golden_dir):
    """ViT-L fp8 engine (e4m3 encoder linears) on the 3840x2160 frame, Full-TAB 4320x3840 and Half-TAB 2160x3840 through
    d2s_pipeline, against the reference's fp32 depth (vitl_r518_4k: CPU branch, ::3 decimation) and the oracle warp.
    Both e4m3 schemes (all four encoder linears / FC1 + FC2 only) are gated on mean AND max against the reference's own model under
    the same operand quantisation (tests/golden/fp8_frontier_vitl_4k.json)."""
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
    from desktop2stereo_amd.weights import make_weights
    from oracle import d2s_oracle as O
    z, meta = _golden(golden_dir, "vitl_r518_4k")
    fr = meta["frames"][0]
    H, W = fr["h"], fr["w"]
    assert (H, W) == (2160, 3840)
    frame = synth.structured_frame(H, W, fr["seed"])
    cfg = MODELS["vitl"]
    h, w, stride = engine_shape(H, W, 518)
    assert stride == 3 and (h, w) == (294, 518)
    wts = make_weights(cfg, 0)
    ref_full = O.upsample_depth(z["f0_post_depth"], H, W)
    ft = _t(frame[None], dev)
    # e4m3 gates (round 5): REFERENCE-DERIVED -- tests/golden/fp8_frontier_vitl_4k.json holds what the reference's
    # own model loses on this very frame when the same linears' operands are quantised the same way (fp8_scheme_study.py --frontier
    # --4k --json: torch float8_e4m3fn emulation, fp32 accumulation); the HIP engine must stay within 1.25 x its mean and 1.5 x its max
    # (the max of a 152 k-pixel map is one pixel: noisier than the mean), for BOTH schemes
    with open(os.path.join(golden_dir, "fp8_frontier_vitl_4k.json")) as f:
        emu = json.load(f)["rows"]
    emu = {"fp8": emu["all four e4m3"], "fp8_mlp": emu["MLP only (FC1 + FC2)"]}
    for prec in ("bf16", "fp8", "fp8_mlp"):
        eng = ops.Engine(cfg, wts, h, w, 1, prec)
        if prec != "bf16":
            eng.calibrate(ops.preprocess(_t(synth.structured_frame(H, W, 0), dev), 518))
        for mode, shape in (("Full-TAB", (4320, 3840, 3)), ("Half-TAB", (2160, 3840, 3))):
            p = PipelineParams(depth_resolution=518, display_mode=mode)
            sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, mode, p.fill_16_9)
            out, depth = eng.pipeline(ft, p, sp, want_depth=True)
            out, depth = out.cpu().numpy()[0], depth.cpu().numpy()[0]
            assert out.shape == shape
            d = np.abs(depth - ref_full)
            lsb_same, frac_same = _warp_vs_oracle(out, frame, depth, mode, p)
            print(f"[config 3, ViT-L {prec}, 4K {mode}] depth vs reference fp32: max {d.max():.4f} mean {d.mean():.5f}; "
                  f"warp (same depth) max {lsb_same} LSB, {frac_same:.2e} > 1 LSB")
            assert lsb_same <= 1, (prec, mode, lsb_same)
            if prec == "bf16":
                # reference-derived (round 4): the reference's own bf16 CPU autocast on this very frame sits max 0.0250 / mean 0.00371 from
                # its fp32 result (tests/golden/vitl_r518_4k_bf16, model resolution; the up-sample is a convex combination)
                zb = np.load(os.path.join(golden_dir, "vitl_r518_4k_bf16.npz"))
                gap = np.abs(zb["f0_post_depth"].astype(np.float32) - z["f0_post_depth"])
                assert d.max() <= gap.max() and d.mean() <= gap.mean(), (d.max(), d.mean(), gap.max(), gap.mean())
            else:
                b = emu[prec]
                print(f"[config 3, {prec}] the reference's model under the same operand quantisation (emulated): mean {b['mean']:.5f} max {b['max']:.4f}")
                assert d.mean() <= 1.25 * b["mean"] and d.max() <= 1.5 * b["max"], (prec, float(d.mean()), float(d.max()), b)
        eng.close()


def test_config5_mixed_64_frames(dev):
    """64 frames, sizes drawn with seed 0 from {1280x720, 1920x1080, 2560x1440} (bench.py --mixed 64 uses the same draw):
    depth.pipeline_mixed (ONE 64-frame model batch, per-size pre-process and warp) == the per-frame path on every frame,
    and == the oracle (fp32 ViT-S engine) on one frame of each size."""
    from desktop2stereo_amd import depth as D, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams
    from desktop2stereo_amd.weights import make_weights
    from oracle import d2s_oracle as O
    sizes = [(720, 1280), (1080, 1920), (1440, 2560)]
    pick = np.random.default_rng(0).integers(0, 3, 64)
    assert len(set(pick.tolist())) == 3
    first = {int(np.argmax(pick == k)) for k in range(3)}                 # the frames the oracle is run on: structured scenes
    frames = [synth.structured_frame(*sizes[k], 9000 + i) if (i in first or i % 8 == 1) else synth.noise_frame(*sizes[k], 9000 + i)
              for i, k in enumerate(pick)]
    p = PipelineParams(depth_resolution=518, display_mode="Full-SBS")
    try:
        # (1) the config's model: ViT-B bf16, batch 64 vs batch 1 (different GEMM tiles, LN kernels vs LN folds: bf16-level noise)
        D.configure("vitb", params=p, precision="bf16", max_batch=64)
        outs = D.pipeline_mixed(frames)
        assert [tuple(o.shape) for o in outs] == [(f.shape[0], 2 * f.shape[1], 3) for f in frames]
        outs = [o.cpu().numpy() for o in outs]
        # bf16 at batch 64 (LN kernels, 128x128 tiles) vs batch 1 (LN folded into the linears, small tiles) are two bf16
        # evaluations of the same frame: depth differs by ~2e-3 mean, i.e. a few hundredths of a pixel of shift -- visible
        # as multi-LSB changes only across sharp edges.  Compared on the structured frames (on noise frames EVERY pixel is
        # an edge); the exactness claim is part (2), fp32.
        worst, fracs = 0, []
        for i in sorted(first | {1, 9, 57}):
            one = D.pipeline(frames[i][None], display_mode="Full-SBS").cpu().numpy()[0]
            diff = np.abs(one.astype(np.int16) - outs[i].astype(np.int16))
            worst = max(worst, int(diff.max())); fracs.append(float((diff > 1).mean()))
        print(f"[config 5, ViT-B bf16] 64-frame mixed batch vs per-frame pipeline (structured frames): max {worst} LSB, "
              f"fraction of bytes > 1 LSB: worst {max(fracs):.2e}, mean {np.mean(fracs):.2e}")
        assert max(fracs) <= 2e-2, fracs
        # (2) parity class: ViT-S fp32, same 64 frames, the oracle on the first frame of each size
        D.configure("vits", params=p, precision="fp32", max_batch=64)
        outs = [o.cpu().numpy() for o in D.pipeline_mixed(frames)]
        cfg = MODELS["vits"]
        orc = O.PipelineOracle(cfg, make_weights(cfg, 0), 518)
        for k in range(3):
            i = int(np.argmax(pick == k))
            f = frames[i]
            d = orc.predict_depth(f)
            want = O.to_u8(orc.make_sbs(f, d, ipd_uv=p.ipd, depth_ratio=p.depth_strength, display_mode="Full-SBS", fill_16_9=p.fill_16_9))
            diff = np.abs(outs[i].astype(np.int16) - want.astype(np.int16))
            print(f"[config 5, ViT-S fp32] frame {i} {f.shape[1]}x{f.shape[0]} vs oracle: max {int(diff.max())} LSB, {(diff > 1).mean():.2e} > 1 LSB")
            assert diff.max() <= 1, (i, diff.max(), (diff > 1).mean())          # measured: 1 LSB, no byte beyond
            one = D.pipeline(f[None], display_mode="Full-SBS").cpu().numpy()[0]
            d1 = np.abs(one.astype(np.int16) - outs[i].astype(np.int16))
            assert d1.max() <= 1 and (d1 > 0).mean() <= 1e-3, (i, d1.max())         # fp32: batch 64 == batch 1 up to summation order
    finally:
        D.configure("tiny", params=PipelineParams(depth_resolution=140), precision="fp32", max_batch=4)


def test_safetensors_round_trip(dev, tmp_path):
    """A real checkpoint file: a random-init HF DepthAnythingForDepthEstimation saved with
    save_pretrained(safe_serialization=True) (what the reference's convert.py:14-24 does) -> configure(weights=path) ->
    the same depth as the same tensors handed over in memory."""
    transformers = pytest.importorskip("transformers")
    from transformers import DepthAnythingConfig, DepthAnythingForDepthEstimation
    from desktop2stereo_amd import depth as D, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams
    cfg = MODELS["tiny"]
    hf = DepthAnythingConfig(
        backbone_config=dict(model_type="dinov2", hidden_size=cfg.hidden, num_attention_heads=cfg.heads,
                             num_hidden_layers=cfg.layers, image_size=518, patch_size=14, out_indices=list(cfg.out_indices),
                             apply_layernorm=True, reshape_hidden_states=False),
        reassemble_hidden_size=cfg.hidden, neck_hidden_sizes=list(cfg.neck), fusion_hidden_size=cfg.fusion,
        head_hidden_size=cfg.head_hidden)
    torch.manual_seed(3)
    m = DepthAnythingForDepthEstimation(hf).eval()
    with torch.no_grad():
        for n, prm in m.named_parameters():                 # HF init leaves LayerScale = 1 and biases = 0: make every tensor matter
            if prm.dim() == 1:
                prm.add_(0.05 * torch.randn_like(prm))
    m.save_pretrained(str(tmp_path), safe_serialization=True)
    path = os.path.join(str(tmp_path), "model.safetensors")
    assert os.path.exists(path)
    sd = {k: v.detach().float().numpy() for k, v in m.state_dict().items() if "mask_token" not in k}
    p = PipelineParams(depth_resolution=140)
    f = synth.structured_frame(270, 480, 12)
    try:
        D.configure(cfg, weights=path, params=p, precision="fp32")
        d_file = D.predict_depth(f, use_temporal_smooth=False).cpu().numpy()
        D.configure(cfg, weights=sd, params=p, precision="fp32")
        d_mem = D.predict_depth(f, use_temporal_smooth=False).cpu().numpy()
        assert np.array_equal(d_file, d_mem)
        # and it is the model HF computes: the reference's forward (model(pixel_values).predicted_depth, depth.py:1778) on the CPU
        from oracle import d2s_oracle as O
        x = O.normalise(O.resize_patch_aligned(np.ascontiguousarray(f.transpose(2, 0, 1)), 140))
        with torch.no_grad():
            raw = m(pixel_values=torch.from_numpy(x)[None]).predicted_depth[0].numpy()
        want = O.upsample_depth(O.post_process_depth(raw, p.foreground_scale, p.aa_strength), 270, 480)
        assert np.abs(d_file - want).max() <= 1e-3, np.abs(d_file - want).max()
    finally:
        D.configure("tiny", params=PipelineParams(depth_resolution=140), precision="fp32", max_batch=4)
