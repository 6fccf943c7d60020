#!/usr/bin/env python
"""Per-kernel instruction counts from the device asm tools/cc_one.sh leaves under /tmp/t: IEEE divisions (v_div_scale pairs),
transcendentals, quarter-rate integer multiplies.  A runtime-uniform `if (flag)` around loop-invariant arithmetic gets hoisted and
speculated by the compiler (round 6: dibr.hip's feather block cost every pixel four divisions and a powf) -- this lists where to look.
    tools/cc_one.sh post zz; python tools/isa_scan.py post"""
import re
import subprocess
import sys

for f in sys.argv[1:]:
    cur, stats = None, {}
    for line in open(f"/tmp/t/{f}.s"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); stats[cur] = dict(n=0, div=0, trans=0, mul=0, scratch=0); continue
        if cur is None:
            continue
        t = line.split()
        if not t:
            continue
        op = t[0]
        s = stats[cur]
        if op.startswith(("v_", "s_", "ds_", "global_", "buffer_", "scratch_")): s["n"] += 1
        if op.startswith("v_div_scale"): s["div"] += 1
        if op.startswith(("v_exp", "v_log", "v_sqrt", "v_rsq", "v_rcp", "v_sin", "v_cos")): s["trans"] += 1
        if op.startswith(("v_mad_u64", "v_mad_i64", "v_mul_hi", "v_mul_lo")): s["mul"] += 1
        if op.startswith("scratch_"): s["scratch"] += 1
        if ".end_amdhsa_kernel" in line: cur = None
    for k, v in stats.items():
        if v["n"] > 50:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0][:90]
            print(f"{f:9s} n={v['n']:5d} div={v['div'] // 2:3d} trans={v['trans']:3d} qmul={v['mul']:3d} scratch={v['scratch']:3d}  {name}")
