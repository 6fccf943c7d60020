#!/usr/bin/env python
"""Rebuild profiles/pmc_traffic.json from rocprofv3 --pmc CSVs in one step.

    tools/pmc_traffic.sh gpurun_out/pmc_r2        # on the GPU box: calibration copy + bench passes (FETCH_SIZE / WRITE_SIZE)
    python tools/pmc_traffic.py gpurun_out/pmc_r2 # here: -> profiles/pmc_traffic.json (+ a markdown table on stdout)

Per kernel class (bench.py's names) and batch: HBM-side bytes per launch = FETCH_SIZE * f_fetch + WRITE_SIZE * f_write,
where the factors come from the calibration kernels of tools/ubench/copy_calib.hip (bytes they are KNOWN to move / what
the counter reported, in the same run).  MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced read on gfx950
(f_fetch ~ 2); WRITE_SIZE is uncalibrated there -- this measures it.
"""
import collections, csv, glob, json, os, sys

CLASSES = [  # (bench.py class, substring(s) of the kernel name)
    ("gemm_linear", ("gemm_pp_kernel", "gemm_glds_kernel", "gemm_sk_kernel", "pp_tail_reduce_kernel")),
    ("stereo_warp", ("stereo_warp",)),
    ("attention", ("attention_kernel", "attention32_kernel")),
]
KNOWN = {"d2s_calib_copy16": (512 << 20, 512 << 20), "d2s_calib_copy16to8": (512 << 20, 256 << 20)}


def read(d, counter):
    """-> {kernel name: [values]} for one counter from every *_counter_collection.csv under d."""
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    root = sys.argv[1]
    out = {"source": root, "calibration": {}, "traffic_bytes_per_launch": {}}
    # provenance: the digest of the kernel sources the passes ran on (written on the GPU box by pmc_traffic.sh), the
    # date of the run, and the commit this file is rebuilt at -- bench.py compares the digest with the tree it runs on
    for key, fn in (("kernel_sources_sha256", "kernel_sources.sha256"), ("date_utc", "date_utc.txt")):
        try:
            out[key] = open(os.path.join(root, fn)).read().strip()
        except OSError:
            out[key] = None
    try:
        import subprocess
        out["commit"] = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__)), text=True).strip()
    except Exception:
        out["commit"] = None
    for B in sorted({os.path.basename(p).split("_b")[-1] for p in glob.glob(os.path.join(root, "bench_b*")) if os.path.isdir(p)}, key=int):
        d = os.path.join(root, f"bench_b{B}")
        fetch, write = read(d, "FETCH_SIZE"), read(d, "WRITE_SIZE")          # counter units are calibrated below, not assumed
        cf, cw = read(os.path.join(root, "calib"), "FETCH_SIZE"), read(os.path.join(root, "calib"), "WRITE_SIZE")
        ff, fw = [], []
        for k, (rb, wb) in KNOWN.items():
            kf = [v for n, vs in cf.items() if n.startswith(k + "(") or n == k for v in vs]
            kw = [v for n, vs in cw.items() if n.startswith(k + "(") or n == k for v in vs]
            if kf: ff.append(rb / (sum(kf) / len(kf)))
            if kw: fw.append(wb / (sum(kw) / len(kw)))
        f_fetch = sum(ff) / len(ff) if ff else None
        f_write = sum(fw) / len(fw) if fw else None
        out["calibration"][B] = {"bytes_per_FETCH_SIZE_unit": f_fetch, "bytes_per_WRITE_SIZE_unit": f_write, "per_kernel_fetch": ff, "per_kernel_write": fw}
        for cls, subs in CLASSES:
            vf = [v for n, vs in fetch.items() if any(s in n for s in subs) for v in vs]
            vw = [v for n, vs in write.items() if any(s in n for s in subs) for v in vs]
            if vf and vw and f_fetch and f_write:
                t = (sum(vf) / len(vf)) * f_fetch + (sum(vw) / len(vw)) * f_write
                out["traffic_bytes_per_launch"].setdefault(cls, {})[B] = t
                print(f"batch {B:>3} {cls:12s}: {len(vf):5d} launches, fetch {sum(vf)/len(vf)*f_fetch/1e6:9.2f} MB + write {sum(vw)/len(vw)*f_write/1e6:9.2f} MB = {t/1e6:9.2f} MB / launch")
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
