"""Time the MJPEG sink (csrc/jpeg.hip) on the device and libjpeg-turbo (Pillow) on the host for the same frames.
    python tools/jpeg_bench.py [--h 1080 --w 3840 --batch 1 --quality 90 --kind scene|noise]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import io
import time

import numpy as np
import torch

from desktop2stereo_amd import ops


def frame(H, W, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    rgb = np.stack([(xx // 15) % 256, (yy // 4) % 256, ((xx + 2 * yy) // 9) % 256], -1).astype(np.int64)
    return np.clip(rgb + rng.integers(-6, 7, rgb.shape), 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=3840)
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 16])
    ap.add_argument("--quality", type=int, nargs="+", default=[90, 100])
    ap.add_argument("--kind", nargs="+", default=["scene", "noise"])
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    for kind in a.kind:
        for q in a.quality:
            for B in a.batch:
                host = np.stack([frame(a.h, a.w, kind, s) for s in range(B)])
                dev = torch.from_numpy(host).cuda()
                out, sizes = ops.jpeg_encode(dev, q)
                torch.cuda.synchronize()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(a.iters):
                    out, sizes = ops.jpeg_encode(dev, q)
                ev1.record()
                torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1) / a.iters
                nbytes = int(sizes.sum())
                line = (f"{kind:5s} q={q:3d} B={B:2d} {a.h}x{a.w}: GPU {ms * 1e3 / B:8.1f} us/frame ({B / ms * 1e3:8.0f} fps, "
                        f"{host.nbytes / ms / 1e6:7.1f} GB/s of RGB in), {nbytes / B / 1e6:5.2f} MB/frame")
                if B == a.batch[0]:
                    from PIL import Image
                    t0 = time.perf_counter()
                    n = 3
                    for i in range(n):
                        buf = io.BytesIO()
                        Image.fromarray(host[0]).save(buf, "JPEG", quality=q, subsampling="4:2:0", optimize=False)
                    cpu = (time.perf_counter() - t0) / n
                    same = buf.getvalue() == out[0, :int(sizes[0])].cpu().numpy().tobytes()
                    line += f" | libjpeg-turbo 1 core {cpu * 1e3:6.1f} ms/frame, identical bytes: {same}"
                print(line, flush=True)


if __name__ == "__main__":
    main()
