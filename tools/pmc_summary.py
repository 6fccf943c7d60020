#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per (kernel, grid size).  usage: pmc_summary.py dir [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt not in k: continue
        key = (k.split("(")[0].replace("unsigned short", "bf16").replace("void d2s::", ""), r.get("Grid_Size", ""))
        a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for key in sorted(acc):
    print(key)
    c = {n: v[0] / v[1] for n, v in acc[key].items()}
    for n in sorted(c): print(f"    {n:32s} {c[n]:16.1f}")
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        w = c["SQ_WAVE_CYCLES"]
        print("    -> wait_any %.2f  wait_inst_any %.2f  active_inst %.2f  (of wave cycles)" % (
            c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        print("    -> mfma_busy / (gui_active*4 simd*256 cu... raw) %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 1024)))
    if "TCC_HIT_sum" in c:
        print("    -> L2 hit rate %.3f" % (c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
