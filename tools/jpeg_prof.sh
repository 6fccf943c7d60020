#!/bin/bash
# MJPEG sink kernels under rocprofv3 --kernel-trace (per-kernel us); run on the GPU box:  tools/jpeg_prof.sh [out_dir] [jpeg_bench args]
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/jpeg_prof}; shift; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o j -- python tools/jpeg_bench.py --quality 90 "$@" > $OUT/j.log 2>&1
python - $OUT/j_results.db <<'P'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for r in cur.execute("select name, grid_x, grid_y, count(*), avg(duration), min(duration) from kernels where name like '%jpeg%' group by 1,2,3 order by 3,1"):
    print(f"{r[0].split('(')[0][-34:]:36s} grid {r[1]:>8} y {r[2]:>3}  calls {r[3]:>4}  avg {r[4] / 1e3:8.2f} us  min {r[5] / 1e3:8.2f} us")
P
