"""Time the REFERENCE ITSELF on BASELINE configs[1]'s workload (build container only: it imports /root/reference through
ref_harness.py and cannot travel to the GPU box).  SURVEY.md section 8d: ViT-B, Depth Resolution 518, 1920x1080 frame,
predict_depth + make_sbs Full-SBS; as shipped (bf16 CPU autocast, depth.py:661-664) with torch.set_num_threads(1)
(depth.py:19) and with all cores; 3 warm-ups (the reference warms up the same way, depth.py:1861), >= 20 timed frames, median.

    python tests/golden/time_reference.py [--frames 20] [--model vitb]   ->  profiles/r4_reference_cpu_timing.json
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import statistics
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--model", default="vitb")
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r4_reference_cpu_timing.json"))
    a = ap.parse_args()
    import torch
    from ref_harness import load_reference
    from desktop2stereo_amd import synth
    D = load_reference(a.model, 518, seed=0, fp32=False)          # as shipped: CPU autocast -> bf16
    ncores = os.cpu_count() or 1
    frames = [synth.noise_frame(1080, 1920, 100 + i) for i in range(4)]
    rows = {}
    for label, threads in (("one_thread_as_shipped", 1), ("all_cores", ncores)):
        torch.set_num_threads(threads)
        D.depth_stabilizer.prev = None
        ts = []
        for i in range(3 + a.frames):
            f = frames[i % len(frames)]
            t0 = time.perf_counter()
            d = D.predict_depth(f, use_temporal_smooth=False)
            D.make_sbs(f, d, ipd_uv=0.064, depth_ratio=4.0, convergence=0.0, fill_16_9=False, display_mode="Full-SBS")
            dt = time.perf_counter() - t0
            if i >= 3:
                ts.append(dt)
        med = statistics.median(ts)
        rows[label] = {"threads": threads, "frames": len(ts), "median_s_per_frame": med, "frames_per_s": 1.0 / med,
                       "min_s": min(ts), "max_s": max(ts)}
        print(label, rows[label], flush=True)
    out = {"what": "the reference's own predict_depth + make_sbs (imported from /root/reference via tests/golden/ref_harness.py), "
                   f"DepthAnything-v2-{a.model} seeded synthetic weights, Depth Resolution 518, 1920x1080 noise frames, Full-SBS, "
                   "bf16 CPU autocast as shipped, EMA off",
           "where": "build container (the reference cannot travel to the GPU box)", "cpu_model": cpu_model(), "host_cpus": ncores,
           "torch": torch.__version__, "warmups": 3, "rows": rows}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
