"""CPU restatement of the reference's GLSL DIBR fragment shader (viewer.py:386-631): the warp with disocclusion
in-painting that the default Viewer / OpenXR modes show.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PINNED (round 5) by tests/golden/dibr.npz: the reference's own shader text compiled and run off-screen here -- OpenGL ES 3.0 on
the SwiftShader software renderer found inside this image's `kaleido` wheel (tests/golden/gl_harness.py, make_golden_dibr.py;
four mechanical ES patches listed there) -- and tests/test_oracle_golden.py holds this file to those renders.  What a reader
comparing with a GL run needs to know:

  * ``u_resolution`` is declared (viewer.py:395) but never assigned anywhere in the reference, so
    ``pixel_size = 1.0 / u_resolution`` (viewer.py:413) is a division by the default 0 -- undefined in the reference
    itself.  This restatement (and the HIP kernel) take the resolution as a parameter, default = source size, which is
    what every ``pixel_size`` use in the shader evidently intends (offsets in texels); the fixtures set the uniform to that.
  * texture() is restated as exact float32 GL_LINEAR filtering with texel centres at (i+0.5)/N and GL_REPEAT wrapping
    (moderngl's defaults; viewer.py:2385-2386 sets no filter / repeat flags); GL implementations filter RGB8 with ~8-bit
    sub-texel weights, so a render differs from this file by up to ~1 level of 255 (measured against SwiftShader: max 0.73,
    mean 0.016 on a scene with hard depth edges).
  * BLENDING: the reference enables GL_BLEND only around its overlay quad (viewer.py:1304-1307) -- the stereo quads are drawn
    with blending OFF, so the window shows ``frag_color.rgb`` as written and ``frag_color.a`` (screen-edge clip, rounded
    corners) lands in the framebuffer's alpha channel, which a desktop window ignores; an alpha-compositing consumer (the
    OpenXR layer) shows rgb * a over what lies behind.  ``dibr_eye(rgba=True)`` returns both, un-multiplied; the default
    return value is rgb * a (composited over black), kept for the round 1-4 tests.

Colours are carried in 0..255 instead of GL's normalised 0..1 (every colour operation is linear).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _tex(img: np.ndarray, u: np.ndarray, v: np.ndarray) -> np.ndarray:
    """GL_LINEAR + GL_REPEAT lookup. img [H,W] or [H,W,C]; u,v float32 arrays (same shape)."""
    H, W = img.shape[:2]
    x = u.astype(F32) * F32(W) - F32(0.5)
    y = v.astype(F32) * F32(H) - F32(0.5)
    x0f, y0f = np.floor(x), np.floor(y)
    fx, fy = (x - x0f).astype(F32), (y - y0f).astype(F32)
    x0 = np.mod(x0f.astype(np.int64), W)
    y0 = np.mod(y0f.astype(np.int64), H)
    x1, y1 = np.mod(x0 + 1, W), np.mod(y0 + 1, H)
    if img.ndim == 3:
        fx, fy = fx[..., None], fy[..., None]
    a = img[y0, x0].astype(F32)
    b = img[y0, x1].astype(F32)
    c = img[y1, x0].astype(F32)
    d = img[y1, x1].astype(F32)
    top = a + (b - a) * fx
    bot = c + (d - c) * fx
    return (top + (bot - top) * fy).astype(F32)


def _smoothstep(e0, e1, x):
    t = np.clip((x - F32(e0)) / F32(e1 - e0), F32(0), F32(1)).astype(F32)
    return (t * t * (F32(3) - F32(2) * t)).astype(F32)


def _inpaint(rgb, dep, u, v, cdi, par, sweep_sign, ps, search_radius, tol, blur):
    """push_pull_inpaint, viewer.py:437-506, for the pixels given by flat arrays u, v, cdi."""
    n = u.shape[0]
    best = np.zeros((n, 3), F32)
    bw = np.zeros(n, F32)
    sx, sy = F32(par[0] * ps[0] * sweep_sign), F32(par[1] * ps[0] * sweep_sign)   # both components use pixel_size.x (:442)
    active = np.ones(n, bool)
    for i in range(1, int(search_radius) + 1):                                     # phase 1 (:445-466)
        su, sv = u + sx * F32(i), v + sy * F32(i)
        ok = active & ~((su < 0) | (sv < 0) | (su > 1) | (sv > 1))
        sdi = F32(1) - _tex(dep, su, sv)
        ok &= sdi > cdi + F32(tol)
        w = (np.exp(F32(-i * 0.15), dtype=F32) * (F32(1) + (sdi - cdi) * F32(10))).astype(F32)
        col = _tex(rgb, su, sv)
        best[ok] += col[ok] * w[ok, None]
        bw[ok] += w[ok]
        active &= ~(ok & (bw > 5))                                                  # early exit (:464)
    need2 = bw < 2                                                                  # phase 2 (:469-481)
    for i in range(1, int(search_radius) + 1):
        su, sv = u - sx * F32(i), v - sy * F32(i)
        ok = need2 & ~((su < 0) | (sv < 0) | (su > 1) | (sv > 1))
        sdi = F32(1) - _tex(dep, su, sv)
        ok &= sdi > cdi + F32(tol)
        w = np.exp(F32(-i * 0.2), dtype=F32)
        col = _tex(rgb, su, sv)
        best[ok] += col[ok] * w
        bw[ok] += w
    out = _tex(rgb, u, v)                                                           # fallback (:505)
    has = bw > F32(0.01)                                                            # phase 3 (:484-502)
    blurred = best / np.maximum(bw, F32(1e-30))[:, None]
    va = blurred * F32(0.5)
    vw = np.full(n, 0.5, F32)
    for dy in (-1, 1):
        vv = v + F32(dy * ps[1] * blur)
        ok = has & (vv >= 0) & (vv <= 1)
        vdi = F32(1) - _tex(dep, u, vv)
        ok &= vdi > cdi + F32(tol * 0.5)
        col = _tex(rgb, u, vv)
        va[ok] += col[ok] * F32(0.25)
        vw[ok] += F32(0.25)
    out[has] = (va / vw[:, None])[has]
    return out.astype(F32)


def dibr_eye(rgb_u8_hwc: np.ndarray, depth: np.ndarray, eye_offset: float, depth_strength: float, convergence: float = 0.0,
             out_h: int = 0, out_w: int = 0, roll: float = 0.0, res=None, search_radius=12.0, tol=0.012, blur=2.5,
             feather=False, feather_width=0.02, corner_radius=0.0, viewport=None, rgba=False) -> np.ndarray:
    """One eye of FRAGMENT_SHADER.main (viewer.py:533-631) rendered into an out_h x out_w viewport -> float32
    [out_h,out_w,3] in 0..255 (colour * alpha over black).  eye_offset: -ipd_uv/2 left, +ipd_uv/2 right
    (viewer.py:2701, 2714); depth_strength = viewer.depth_strength (0.1) * depth_ratio (viewer.py:1334, 2686)."""
    H, W = depth.shape
    oh, ow = out_h or H, out_w or W
    rw, rh = res or (W, H)
    ps = (F32(1) / F32(rw), F32(1) / F32(rh))
    rgb = rgb_u8_hwc.astype(F32)
    dep = depth.astype(F32)
    v, u = np.meshgrid((np.arange(oh, dtype=F32) + F32(0.5)) / F32(oh), (np.arange(ow, dtype=F32) + F32(0.5)) / F32(ow),
                       indexing="ij")
    c, s = F32(np.cos(roll)), F32(np.sin(roll))
    sg = F32(np.sign(eye_offset))
    par = (c * sg, s * sg)                                                          # :540
    sweep_sign = -1.0 if eye_offset > 0 else 1.0                                    # :541
    dsx, dsy = F32(par[0] * ps[0] * F32(1.5)), F32(par[1] * ps[1] * F32(1.5))       # :545
    d0 = _tex(dep, u, v)
    dm = _tex(dep, u - dsx, v - dsy)
    dp = _tex(dep, u + dsx, v + dsy)
    d = (d0 * F32(0.7) + dm * F32(0.15) + dp * F32(0.15)).astype(F32)              # :549
    dinv = -d
    shaped = dinv * (F32(1) + F32(0.35) * (F32(1) - d))                             # :554
    shift = shaped + F32(convergence)
    fall = _smoothstep(0.0, 0.05, u) * _smoothstep(1.0, 0.95, u)                    # :560-562
    px = (F32(eye_offset) * shift * F32(depth_strength) * fall).astype(F32)        # :563
    su, sv = (u - px * c).astype(F32), (v - px * s).astype(F32)                     # :564
    oob = (su < 0) | (su > 1) | (sv < 0) | (sv > 1)                                 # :422-425
    s2x, s2y = F32(par[0] * ps[0] * F32(2)), F32(par[1] * ps[1] * F32(2))           # :428
    jump = np.abs(_tex(dep, u - s2x, v - s2y) - _tex(dep, u + s2x, v + s2y))
    conf = np.where(oob, F32(1), _smoothstep(0.04, 0.10, jump)).astype(F32)         # :434
    color = _tex(rgb, su, sv)                                                       # :570
    m = conf > F32(0.001)
    if m.any():
        filled = _inpaint(rgb, dep, u[m], v[m], dinv[m], par, sweep_sign, ps, search_radius, tol, blur)
        cm = conf[m][:, None]
        color[m] = color[m] * (F32(1) - cm) + filled * cm                           # mix (:575)
    bx = _smoothstep(-0.001, 0.001, su) * _smoothstep(1.001, 0.999, su)             # :582
    by = _smoothstep(-0.001, 0.001, sv) * _smoothstep(1.001, 0.999, sv)
    alpha = np.minimum(bx, by)
    if feather:                                                                      # :587-616 (viewport uv, y up)
        vx, vy, vw_, vh_ = viewport if viewport is not None and viewport[2] > 0 else (0.0, 0.0, float(ow), float(oh))
        xs = np.arange(ow, dtype=F32)[None, :] + F32(0.5)                            # gl_FragCoord: pixel centres, y up
        ys = F32(oh) - (np.arange(oh, dtype=F32)[:, None] + F32(0.5))
        fu = np.broadcast_to((xs - F32(vx)) / F32(vw_), (oh, ow)).astype(F32)
        fv = np.broadcast_to((ys - F32(vy)) / F32(vh_), (oh, ow)).astype(F32)
        fw = F32(feather_width)
        fo = (_smoothstep(0.0, fw, fu) * _smoothstep(0.0, fw, F32(1) - fu) * _smoothstep(0.0, fw, fv)
              * _smoothstep(0.0, fw, F32(1) - fv))
        color = color * np.power(fo, F32(0.7))[..., None]
    if corner_radius > 0:                                                            # :617-624 (quad uv, Inigo Quilez rounded box)
        r = F32(corner_radius)
        dx, dy = np.abs(u - F32(0.5)) - F32(0.5) + r, np.abs(v - F32(0.5)) - F32(0.5) + r
        sdf = np.sqrt(np.maximum(dx, 0) ** 2 + np.maximum(dy, 0) ** 2).astype(F32) + np.minimum(np.maximum(dx, dy), 0) - r
        alpha = np.minimum(alpha, F32(1) - _smoothstep(0.0, 0.01, sdf.astype(F32)))
    if rgba:        # frag_color as the shader writes it: rgb (0..255 here) and alpha, un-multiplied
        return np.concatenate([color.astype(F32), alpha[..., None].astype(F32)], -1)
    return (color * alpha[..., None]).astype(F32)


def dibr_sbs(rgb_u8_hwc, depth, ipd_uv=0.064, depth_ratio=1.0, convergence=0.0, display_mode="Full-SBS",
             viewer_depth_strength=0.1, alpha="window", **kw) -> np.ndarray:
    """Both eyes packed like the viewer lays out its viewports for an undistorted window (viewer.py:2688-2830):
    Full-SBS [H,2W], Half-SBS [H,W] (each eye W/2 columns), Full-TAB [2H,W], Half-TAB [H,W] (each eye H/2 rows).
    alpha: "window" = frag_color.rgb as the reference's window shows it (blending off), "premultiplied" = rgb * a,
    "rgba" = four channels (rgb 0..255, a 0..1)."""
    H, W = depth.shape
    eh = H // 2 if display_mode == "Half-TAB" else H
    ew = W // 2 if display_mode == "Half-SBS" else W
    ds = viewer_depth_strength * depth_ratio
    left = dibr_eye(rgb_u8_hwc, depth, -ipd_uv / 2.0, ds, convergence, eh, ew, rgba=True, **kw)
    right = dibr_eye(rgb_u8_hwc, depth, ipd_uv / 2.0, ds, convergence, eh, ew, rgba=True, **kw)
    out = np.concatenate([left, right], 1 if display_mode.endswith("SBS") else 0)
    if alpha == "rgba":
        return out
    return (out[..., :3] * out[..., 3:4]).astype(F32) if alpha == "premultiplied" else np.ascontiguousarray(out[..., :3])
