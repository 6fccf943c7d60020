"""Golden vectors for the viewer's DIBR warp: the REFERENCE's own fragment shader (viewer.py:386-631), read
from /root/reference at generation time, compiled as OpenGL ES 3.0 and run off-screen on SwiftShader (gl_harness.py) in the build
container.

    python tests/golden/make_golden_dibr.py        # -> tests/golden/dibr.npz + dibr.json

Each case renders both eyes exactly as the viewer does (viewer.py:2686-2720: one viewport per eye, u_eye_offset = -/+ ipd_uv / 2,
u_depth_strength = viewer.depth_strength (0.1) * depth_ratio, the full-screen TRIANGLE_STRIP quad, blending off) into an RGBA32F
target and stores frag_color un-multiplied: rgb * 255 * 256 as uint16 fixed point, alpha * 65535 as uint16 (every `row_stride`-th
row).  Inputs are regenerated from seeds (desktop2stereo_amd.synth + the `scene` recipe below); nothing of the shader text is kept.
The one uniform the reference never assigns, u_resolution (pixel_size = 1 / 0, viewer.py:413), is set to the source size -- except in the
two `as_shipped_*` cases, which leave it at (0, 0) and record what the rasteriser makes of the non-finite tap coordinates.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def scene(h, w, seed, kind):
    from desktop2stereo_amd import synth
    return synth.dibr_scene(h, w, seed, kind)


# (name, h, w, seed, scene kind, row stride of stored outputs, per-eye viewport (w, h) or None = source size, uniforms)
CASES = [
    ("small_boxes", 90, 160, 3, "boxes", 1, None, dict(ipd_uv=0.064, depth_ratio=4.0, convergence=0.0)),
    ("small_conv", 96, 128, 4, "boxes", 1, None, dict(ipd_uv=0.064, depth_ratio=2.0, convergence=0.35)),
    ("small_roll", 90, 160, 5, "boxes", 1, None, dict(ipd_uv=0.064, depth_ratio=4.0, convergence=0.0, roll=0.2)),
    ("small_feather", 90, 160, 6, "boxes", 1, None, dict(ipd_uv=0.064, depth_ratio=4.0, convergence=0.1, feather=True, feather_width=0.08,
                                                           corner_radius=0.06)),
    ("small_half", 90, 160, 7, "boxes", 1, (80, 90), dict(ipd_uv=0.064, depth_ratio=4.0, convergence=0.0)),        # Half-SBS eye viewport
    ("small_smooth", 72, 128, 8, "smooth", 1, None, dict(ipd_uv=0.064, depth_ratio=1.0, convergence=0.0)),
    ("hd_boxes", 1080, 1920, 9, "boxes", 45, None, dict(ipd_uv=0.064, depth_ratio=4.0, convergence=0.0)),
    ("hd_half_tab", 1080, 1920, 10, "boxes", 45, (1920, 540), dict(ipd_uv=0.064, depth_ratio=2.0, convergence=0.2)),
    # round 6: a Half-mode viewport with roll, and BASELINE configs[2]'s frame size
    ("small_half_roll", 90, 160, 11, "boxes", 1, (80, 90), dict(ipd_uv=0.064, depth_ratio=4.0, convergence=0.05, roll=-0.15)),
    ("uhd_boxes", 2160, 3840, 12, "boxes", 120, None, dict(ipd_uv=0.064, depth_ratio=4.0, convergence=0.0)),
    # The reference AS SHIPPED: u_resolution is declared (viewer.py:395) and never assigned, so pixel_size = 1.0 / vec2(0) (viewer.py:413)
    # = +inf and every tap at uv +- k * pixel_size has a non-finite coordinate, whose sampling OpenGL leaves undefined.  These two cases
    # record what THIS rasteriser (SwiftShader) returns for that uniform state; they do not pin the kernel (d2s_dibr_params.res_w = 0
    # means "the source size": the shader's evident intent, include/d2s.h) -- tests print the distance, they do not gate on it.
    ("as_shipped_boxes", 90, 160, 3, "boxes", 1, None, dict(ipd_uv=0.064, depth_ratio=4.0, convergence=0.0, as_shipped=True)),
    ("as_shipped_smooth", 72, 128, 8, "smooth", 1, None, dict(ipd_uv=0.064, depth_ratio=1.0, convergence=0.0, as_shipped=True)),
]


def main():
    import gl_harness as G
    (vs, _, _), (fs, l0, l1) = G.reference_shaders()
    vs2, _ = G.to_es300(vs)
    fs2, defaults = G.to_es300(fs)
    gl = G.Gles()
    prog = gl.program(vs2, fs2)
    data, meta = {}, {"cases": [], "gl": {"version": gl.version, "renderer": gl.renderer},
                      "shader": f"/root/reference/viewer.py:{l0}-{l1} (FRAGMENT_SHADER), ES 3.00 patches: see gl_harness.py",
                      "uniform_defaults_from_the_shader_text": defaults,
                      "encoding": "<case>_<eye>_rgb = uint16 rint(frag_color.rgb * 255 * 256); <case>_<eye>_a = uint16 rint(frag_color.a * 65535)"}
    for name, h, w, seed, kind, rs, vp, u in CASES:
        img, dep = scene(h, w, seed, kind)
        tc, td = gl.texture(img, 0), gl.texture(dep, 1)
        ow, oh = vp or (w, h)
        for eye, sign in (("left", -1.0), ("right", 1.0)):
            uni = dict(tex_color=0, tex_depth=1, u_resolution=(0.0, 0.0) if u.get("as_shipped") else (float(w), float(h)), u_eye_offset=float(sign * u["ipd_uv"] / 2.0),
                       u_depth_strength=float(0.1 * u["depth_ratio"]), u_convergence=float(u["convergence"]), u_roll=float(u.get("roll", 0.0)),
                       u_feather_enabled=int(bool(u.get("feather", False))), u_feather_width=float(u.get("feather_width", 0.02)),
                       u_viewport=(0.0, 0.0, float(ow), float(oh)), **{k: float(v) for k, v in defaults.items()})
            if "corner_radius" in u:
                uni["u_corner_radius"] = float(u["corner_radius"])
            out = gl.render(prog, uni, ow, oh)
            assert np.isfinite(out).all()
            data[f"{name}_{eye}_rgb"] = np.rint(np.clip(out[::rs, :, :3], 0, 1) * (255.0 * 256.0)).astype(np.uint16)
            data[f"{name}_{eye}_a"] = np.rint(np.clip(out[::rs, :, 3], 0, 1) * 65535.0).astype(np.uint16)
        gl.delete_texture(tc)
        gl.delete_texture(td)
        meta["cases"].append(dict(name=name, h=h, w=w, seed=seed, scene=kind, row_stride=rs, eye_w=ow, eye_h=oh, **u))
        print("rendered", name, (oh, ow))
    np.savez_compressed(os.path.join(HERE, "dibr.npz"), **data)
    with open(os.path.join(HERE, "dibr.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote dibr", len(data), "arrays")


if __name__ == "__main__":
    main()
