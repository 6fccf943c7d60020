#!/bin/bash
# stereo_warp_gather at batch 1 (kernel time by rocprofv3) against D2S_WARP_WPC = waves per CU the launcher sizes the row bands for.
# run on the GPU box:  tools/warp_wpc_sweep.sh [out_dir]
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m ${1:-$R/gpurun_out/warp_wpc}); mkdir -p $OUT
for w in 4 8 12 16 24 32 48 64; do
  D2S_WARP_WPC=$w timeout 200 rocprofv3 --kernel-trace -d $OUT -o w$w -- python $R/tools/warp_ab.py --batches 1 2 --modes Full-SBS --n 30 > $OUT/w$w.log 2>&1
  python - $OUT/w${w}_results.db $w <<'P'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select grid_x, count(*), avg(duration), min(duration) from kernels where name like '%stereo_warp_gather%' group by 1 order by 1"))
print("WPC", sys.argv[2], "  ".join(f"grid {g}: avg {a / 1e3:.2f} us min {m / 1e3:.2f} ({n})" for g, n, a, m in rows))
P
  rm -f $OUT/w${w}_results.db
done
