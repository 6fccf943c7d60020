"""In-tree build of libd2s_hip.so (hipcc, gfx950 only).

    python -m desktop2stereo_amd.build [--force] [--verbose]

Objects are cached by source mtime under csrc/_build/; the shared library lands next to this file
so it travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libd2s_hip.so")
SOURCES = ["core.cpp", "present.cpp", "ingest.hip", "frame_ops.hip", "dibr.hip", "jpeg.hip", "post.hip", "gemm.hip", "conv3.hip", "gemm_pp.hip", "gemm_sk.hip", "vit_ops.hip", "attention.hip", "temporal.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         # kernarg preload (gfx950): the first 16 argument dwords of a kernel arrive in SGPRs with the wave instead of through a cold
         # s_load -- scalar / pointer arguments up to the first by-value struct.  gemm_glds_kernel's argument order is built around it
         # (gemm.hip); the small kernels (attention, LayerNorm, up-sample, post-process) get all their arguments that way.
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]
FLAGS += os.environ.get("D2S_HIPCC_DEFS", "").split()          # tuning aids only (e.g. -DD2S_PP_TIMING); rebuild with --force
# no FMA contraction in the frame-side / post-process kernels: keeps their float32 op sequence
# comparable with the oracle's (the matrix kernels keep the default fast contraction)
EXTRA = {"frame_ops.hip": ["-ffp-contract=off"], "post.hip": ["-ffp-contract=off"], "ingest.hip": ["-ffp-contract=off"],
         "dibr.hip": ["-ffp-contract=off"],
         # the softmax never produces a NaN (masked scores are -1e30, exp2 of them is 0): without IEEE mode the compiler
         # drops the canonicalising v_max_f32 x, x it otherwise puts in front of every fmaxf on an MFMA result
         # (30 of ~200 VALU instructions per key tile of the batched kernel, which is VALU-bound)
         "attention.hip": ["-fno-honor-nans", "-mno-amdgpu-ieee"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps_mtime() -> float:
    t = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def kernel_sources_digest() -> str:
    """sha256 over the kernel / C-ABI sources (csrc/*.hip, *.cpp, *.h + include/*.h, names and contents, sorted): the identity of the
    code a measurement was taken on.  profiles/pmc_traffic.json records it (tools/pmc_traffic.sh on the GPU box) and bench.py refuses
    that file's traffic figures when the tree it runs on has a different digest."""
    import hashlib
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".cpp", ".h")):
                h.update(f.encode())
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr = _deps_mtime()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            cmd = [hipcc] + FLAGS + EXTRA.get(s, []) + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    if "--digest" in sys.argv:
        print(kernel_sources_digest())
    else:
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
