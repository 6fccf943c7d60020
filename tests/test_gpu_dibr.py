"""GPU: the DIBR warp with disocclusion in-painting (reference viewer.py:386-631).

PINNED (round 5): tests/golden/dibr.npz holds renders of the REFERENCE's own fragment shader, compiled as OpenGL ES 3.0 and run
off-screen on SwiftShader in the build container (tests/golden/gl_harness.py, make_golden_dibr.py: both eyes, hard depth edges,
convergence, roll, feathering + rounded corners, Half viewports, 1080p).  test_dibr_matches_reference_shader_renders holds the HIP
kernel to those renders (rgb and alpha separately); tests/test_oracle_golden.py holds the CPU restatement to them; the remaining
tests compare the kernel with the restatement on parameter sweeps the fixtures do not cover.

Tolerances.  Against a GL render: GL_LINEAR on an RGB8 texture is specified with limited sub-texel weight precision (8 bits is
what GPUs and SwiftShader implement): up to 255/512 of a level per lerp axis, so <= 1 level; the shader is also full of hard
thresholds (depth tests, the `best_weight > 5` early exit, conf > 0.001), so at 1920 columns a 1-ulp difference in a coordinate
flips isolated pixels (measured: 3e-4..6e-4 of the values beyond 1 level, restatement vs render).  Against the restatement (same
float32 filtering on both sides): >= 99.9 % of the values within 0.02 of a level, mean <= 2e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no ROCm device is visible")
    return torch.device("cuda", 0)


def _scene(H, W, seed):
    from desktop2stereo_amd import synth
    img = synth.structured_frame(H, W, seed)
    dep = synth.smooth_depth(H, W, seed).copy()
    dep[H // 5: H // 5 * 3, W // 4: W // 2] = 0.93                 # a near box: sharp edges -> disocclusions
    dep[H // 2: H - H // 6, W // 8 * 5: W // 8 * 6] = 0.05         # and a far slot
    return img, dep.astype(np.float32)


def _check(got, want, what):
    d = np.abs(got.astype(np.float32) - want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert (d <= 0.02).mean() >= 0.999 and d.mean() <= 2e-3, (what, float((d > 0.02).mean()), float(d.mean()), float(d.max()))


@pytest.mark.parametrize("mode", ["Full-SBS", "Half-SBS", "Full-TAB", "Half-TAB"])
def test_dibr_modes_vs_oracle(dev, mode):
    from desktop2stereo_amd import ops
    from oracle import dibr_oracle as R
    img, dep = _scene(180, 320, 1)
    dp = ops.dibr_params(0.064, 4.0, 0.0, mode)
    got = ops.dibr_warp(torch.from_numpy(img).to(dev), torch.from_numpy(dep).to(dev), dp, out_u8=False).cpu().numpy()
    want = R.dibr_sbs(img, dep, 0.064, 4.0, 0.0, mode)
    _check(got, want, mode)
    u8 = ops.dibr_warp(torch.from_numpy(img).to(dev), torch.from_numpy(dep).to(dev), dp).cpu().numpy()
    lsb = np.abs(u8.astype(int) - np.clip(np.rint(want), 0, 255).astype(int))
    assert (lsb <= 1).mean() >= 0.999, (mode, float((lsb > 1).mean()))
    if mode == "Full-SBS":
        assert np.abs(got[:, :320] - img).max() > 1 and np.abs(got[:, :320] - got[:, 320:]).max() > 1    # it does warp


def test_dibr_parameters_vs_oracle(dev):
    """roll, convergence, feathering, explicit u_resolution, large strength (out-of-bounds parallax), batch."""
    from desktop2stereo_amd import ops
    from oracle import dibr_oracle as R
    img, dep = _scene(150, 260, 2)
    ti, td = torch.from_numpy(img).to(dev), torch.from_numpy(dep).to(dev)
    cases = [dict(roll=0.3), dict(convergence=0.5), dict(feather=True), dict(resolution=(520.0, 300.0)),
             dict(depth_ratio=30.0), dict(search_radius=5.0, depth_tolerance=0.3, blur_radius=1.0),
             dict(corner_radius=0.03),                                        # the OpenXR screen's rounded corners (viewer.py:617-624)
             dict(corner_radius=0.2, feather=True, feather_width=0.1, viewport=(10.0, 5.0, 230.0, 140.0))]   # u_viewport != the quad
    for kw in cases:
        okw = dict(kw)
        dr = okw.pop("depth_ratio", 4.0)
        conv = okw.pop("convergence", 0.0)
        am = "premultiplied" if "corner_radius" in okw else "window"         # (alpha < 1 only matters where the SDF / edge clip bite)
        dp = ops.dibr_params(0.064, dr, conv, "Full-SBS", alpha=am, **okw)
        got = ops.dibr_warp(ti, td, dp, out_u8=False).cpu().numpy()
        rk = {}
        if "roll" in okw: rk["roll"] = okw["roll"]
        if "feather" in okw: rk["feather"] = True
        for k in ("feather_width", "corner_radius", "viewport"):
            if k in okw: rk[k] = okw[k]
        if "resolution" in okw: rk["res"] = okw["resolution"]
        if "search_radius" in okw: rk.update(search_radius=5.0, tol=0.3, blur=1.0)
        want = R.dibr_sbs(img, dep, 0.064, dr, conv, "Full-SBS", alpha=am, **rk)
        _check(got, want, kw)
        if am == "premultiplied":                                             # and frag_color itself, four channels
            dp4 = ops.dibr_params(0.064, dr, conv, "Full-SBS", alpha="rgba", **okw)
            got4 = ops.dibr_warp(ti, td, dp4, out_u8=False).cpu().numpy()
            want4 = R.dibr_sbs(img, dep, 0.064, dr, conv, "Full-SBS", alpha="rgba", **rk)
            assert got4.shape == want4.shape and got4.shape[-1] == 4
            _check(got4[..., :3], want4[..., :3], (kw, "rgba rgb"))
            assert np.abs(got4[..., 3] - want4[..., 3]).max() <= 1e-4 and want4[..., 3].min() < 0.5
    img2, dep2 = _scene(150, 260, 3)
    dp = ops.dibr_params(0.064, 4.0, 0.0, "Half-SBS")
    got = ops.dibr_warp(torch.from_numpy(np.stack([img, img2])).to(dev), torch.from_numpy(np.stack([dep, dep2])).to(dev), dp,
                        out_u8=False).cpu().numpy()
    _check(got[0], R.dibr_sbs(img, dep, 0.064, 4.0, 0.0, "Half-SBS"), "batch0")
    _check(got[1], R.dibr_sbs(img2, dep2, 0.064, 4.0, 0.0, "Half-SBS"), "batch1")


def test_to_stereo_inpaint_surface(dev):
    """north_star's convert.to_stereo(..., inpaint=True): same call surface as make_sbs, shader warp underneath."""
    from desktop2stereo_amd import depth as D, _lib
    from oracle import dibr_oracle as R
    img, dep = _scene(120, 200, 4)
    out = D.to_stereo(img, torch.from_numpy(dep), depth_ratio=3.0, display_mode="Full-TAB", inpaint=True)
    assert isinstance(out, np.ndarray) and out.dtype == np.float32 and out.shape == (240, 200, 3)
    _check(out, R.dibr_sbs(img, dep, 0.064, 3.0, 0.0, "Full-TAB"), "to_stereo")
    chw = torch.from_numpy(img).permute(2, 0, 1).float()
    out2 = D.make_sbs(chw, dep, depth_ratio=3.0, display_mode="Full-TAB", inpaint=True)
    assert np.array_equal(out, out2)
    with pytest.raises(_lib.D2SError):
        D.make_sbs(img, dep, inpaint=True, fill_16_9=True)
    with pytest.raises(ValueError):
        D.make_sbs(img, dep, inpaint=True, display_mode="Anaglyph")


def test_dibr_matches_reference_shader_renders(dev, golden_dir):
    """The HIP kernel against renders of the reference's OWN shader (tests/golden/dibr.npz, see the module docstring): per eye,
    frag_color.rgb (0..255) and frag_color.a compared separately through alpha_mode = RGBA.  Small cases: every value within 1
    level, alpha within 1e-3; 1080p cases (every 45th row): >= 99.9 % of the values within 1 level, mean <= 0.06, alpha within 1e-3
    -- the same figures the CPU restatement meets against these renders (tests/test_oracle_golden.py)."""
    import json
    import os
    from desktop2stereo_amd import ops, synth
    z = np.load(os.path.join(golden_dir, "dibr.npz"))
    meta = json.load(open(os.path.join(golden_dir, "dibr.json")))
    for c in meta["cases"]:
        img, dep = synth.dibr_scene(c["h"], c["w"], c["seed"], c["scene"])
        mode = "Full-SBS"
        if (c["eye_w"], c["eye_h"]) == (c["w"] // 2, c["h"]):
            mode = "Half-SBS"
        elif (c["eye_w"], c["eye_h"]) == (c["w"], c["h"] // 2):
            mode = "Half-TAB"
        else:
            assert (c["eye_w"], c["eye_h"]) == (c["w"], c["h"])
        dp = ops.dibr_params(c["ipd_uv"], c["depth_ratio"], c["convergence"], mode, roll=c.get("roll", 0.0), feather=c.get("feather", False),
                             feather_width=c.get("feather_width", 0.02), corner_radius=c.get("corner_radius", 0.0), alpha="rgba")
        got = ops.dibr_warp(torch.from_numpy(img).to(dev), torch.from_numpy(dep).to(dev), dp, out_u8=False).cpu().numpy()
        eyes = (got[:, :c["eye_w"]], got[:, c["eye_w"]:]) if mode.endswith("SBS") else (got[:c["eye_h"]], got[c["eye_h"]:])
        for eye, g in zip(("left", "right"), eyes):
            g = g[::c["row_stride"]]
            rgb = z[f"{c['name']}_{eye}_rgb"].astype(np.float32) / 256.0
            a = z[f"{c['name']}_{eye}_a"].astype(np.float32) / 65535.0
            d = np.abs(g[..., :3] - rgb)
            da = np.abs(g[..., 3] - a)
            print(f"[dibr vs the reference shader's render, {c['name']} {eye}] rgb max {d.max():.3f} mean {d.mean():.4f} "
                  f"{(d > 1).mean():.2e} of values > 1 level ({int((d > 1).sum())} of {d.size}) | alpha max diff {da.max():.1e} (min alpha {a.min():.3f})")
            if c.get("as_shipped"):          # u_resolution left at (0, 0) as the reference ships it: undefined sampling, recorded not gated
                continue
            assert da.max() <= 1e-3, (c["name"], eye, float(da.max()))
            if c["w"] <= 320:
                assert d.max() <= 1.0, (c["name"], eye, float(d.max()))
            else:
                assert (d <= 1.0).mean() >= 0.999 and d.mean() <= 0.06, (c["name"], eye, float((d > 1).mean()), float(d.mean()))


def test_dibr_row_kernels_are_bit_identical_to_the_gather_kernel(dev, monkeypatch):
    """roll == 0 (the desktop viewer) has two faster paths: the gather kernel forming the y half of every texture tap once per pixel
    (row_ctx), and the LDS-window kernel (a block stages its two texture rows once, both eyes; taps are LDS reads at a window index,
    taps beyond the window fall back to the row gather; 256, 512 or 1 024 output columns per block; the five shift-independent depth
    taps of a column shared by its two eyes; feather / corner code compiled out when both are off).  Same expressions on the same operands, so all three must agree bit for bit --
    float32 output, all four modes, hard depth edges (the in-painting's sweeps), convergence, feathering + rounded corners, a
    non-default u_resolution, rgba, a strong parallax, and a depth map beyond 0..1 (shifts past the window margin: the fallback)."""
    from desktop2stereo_amd import ops, synth
    keys = ("D2S_DIBR_NO_ROLL0", "D2S_DIBR_NO_ROWS", "D2S_DIBR_COLS")
    try:
        for (H, W, seed, dscale, kw) in [(270, 480, 3, 1.0, {}), (180, 322, 4, 1.0, dict(convergence=0.03, feather=True, corner_radius=0.03)),
                                         (1080, 1920, 5, 1.0, dict(depth_ratio=3.0)), (200, 300, 6, 1.0, dict(resolution=(640.0, 360.0), alpha="rgba")),
                                         (120, 700, 7, 4.0, dict(depth_ratio=5.0, ipd_uv=0.2))]:
            img, dep = synth.dibr_scene(H, W, seed, "boxes")
            f, d = torch.from_numpy(img).to(dev), torch.from_numpy(dep * np.float32(dscale)).to(dev)
            for mode in ("Full-SBS", "Half-SBS", "Full-TAB", "Half-TAB"):
                dp = ops.dibr_params(display_mode=mode, **kw)
                outs = {}
                for name, env in (("rows", {}), ("rows_256", {"D2S_DIBR_COLS": "256"}), ("rows_1024", {"D2S_DIBR_COLS": "1024"}),
                                  ("row_gather", {"D2S_DIBR_NO_ROWS": "1"}), ("general", {"D2S_DIBR_NO_ROLL0": "1"})):
                    for k in keys:
                        monkeypatch.delenv(k, raising=False)
                    for k, v in env.items():
                        monkeypatch.setenv(k, v)
                    ops.reload_env()
                    outs[name] = ops.dibr_warp(f, d, dp, out_u8=False).cpu().numpy()
                for name in ("rows", "rows_256", "rows_1024", "row_gather"):
                    assert np.array_equal(outs[name], outs["general"]), (H, W, mode, kw, name, float(np.abs(outs[name] - outs["general"]).max()))
                u8 = ops.dibr_warp(f, d, dp).cpu().numpy()              # (env: general) and the uint8 store of the window kernel
                for k in keys:
                    monkeypatch.delenv(k, raising=False)
                ops.reload_env()
                assert np.array_equal(ops.dibr_warp(f, d, dp).cpu().numpy(), u8), (H, W, mode, "uint8")
    finally:
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        ops.reload_env()
