#!/bin/bash
# DIBR kernels under rocprofv3 --kernel-trace (per-kernel us, both entry points); run on the GPU box:  tools/dibr_prof.sh [out_dir]
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/dibr_prof}; mkdir -p $OUT
for v in one; do
  a=""
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o $v -- python tools/dibr_bench.py --batch 1 8 $a > $OUT/$v.log 2>&1
  python - $OUT/${v}_results.db <<'P'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for r in cur.execute("select name, grid_x, grid_z, count(*), avg(duration), min(duration) from kernels where name like '%dibr%' group by 1,2,3 order by 3,1"):
    print(f"{r[0].split('(')[0][-40:]:42s} grid {r[1]:>7} z {r[2]:>2}  calls {r[3]:>3}  avg {r[4] / 1e3:8.2f} us  min {r[5] / 1e3:8.2f} us")
P
done
