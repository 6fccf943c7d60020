"""Golden vectors for the MJPEG sink: run HERE (needs Pillow), commit the output.

The reference encodes with cv2.imencode (reference streamer.py:252, 290), i.e. OpenCV's bundled libjpeg-turbo at
libjpeg's defaults.  cv2 is not installed in this image; Pillow links libjpeg-turbo as well and, with
`subsampling='4:2:0', optimize=False`, issues the same jpeg_set_defaults / jpeg_set_quality(q, TRUE) calls.
Each case = (H, W, quality, kind, seed) -> the exact bytes that library wrote.  Inputs are regenerated from the
seed by `jpeg_case_input` below (also imported by the tests), so only the JPEG bytes are stored.

    python tests/golden/make_jpeg_golden.py
"""
import io
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# sizes exercise: exact MCU multiples, odd sizes (edge replication), W with a dummy luma block column (W%16 in 1..8),
# H with a dummy luma block row (H%16 in 1..8), and H even but not a multiple of 16 — the 1080-row case, where the last
# chroma row is repeated rather than re-derived.
CASES = [(16, 16, 90, "noise"), (32, 48, 100, "scene"), (17, 33, 75, "scene"), (24, 40, 90, "noise"), (9, 9, 50, "scene"),
         (41, 57, 100, "noise"), (64, 100, 90, "scene"), (8, 24, 30, "noise"), (30, 16, 5, "scene"), (40, 72, 95, "scene"),
         (120, 136, 90, "scene"), (56, 64, 1, "noise")]


def jpeg_case_input(H, W, kind, seed):
    r = np.random.default_rng(seed)
    if kind == "noise":
        return r.integers(0, 256, (H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([xx * 255 // max(W - 1, 1), yy * 255 // max(H - 1, 1), (xx + yy) * 3 % 256], -1).astype(np.float64)
    img += r.normal(0, 12, (H, W, 3))
    for _ in range(4):
        y0, x0 = int(r.integers(0, H)), int(r.integers(0, W))
        img[y0:y0 + max(H // 3, 1), x0:x0 + max(W // 3, 1)] = r.integers(0, 256, 3)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    import PIL
    from PIL import Image, features
    out, meta = {}, {"pillow": PIL.__version__, "libjpeg_turbo": bool(features.check_feature("libjpeg_turbo")),
                     "jpeglib": features.version("jpg"), "cases": []}
    for i, (H, W, q, kind) in enumerate(CASES):
        rgb = jpeg_case_input(H, W, kind, 1000 + i)
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, "JPEG", quality=q, subsampling="4:2:0", optimize=False)
        out[f"jpeg_{i}"] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
        meta["cases"].append({"H": H, "W": W, "quality": q, "kind": kind, "seed": 1000 + i, "bytes": len(buf.getvalue())})
    np.savez_compressed(os.path.join(HERE, "jpeg_kat.npz"), **out)
    with open(os.path.join(HERE, "jpeg_kat.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(CASES), "cases,", sum(c["bytes"] for c in meta["cases"]), "JPEG bytes")


if __name__ == "__main__":
    main()
