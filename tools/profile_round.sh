#!/bin/bash
# One command for a round's committed evidence (GPU box):  tools/profile_round.sh <outdir under gpurun_out>
#   1. rocprofv3 --kernel-trace --stats of the default `python bench.py` (batch 1 + batched 32 + tile fit) -> <out>/kernel_trace (rocpd db + csv)
#   2. the bench JSON line of that same run -> <out>/bench_default.json
#   3. PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass, no tracing flags) -> <out>/pmc  (tools/pmc_traffic.sh)
# then here:  python tools/rocprof_summary.py <db> > profiles/rN_..._kernel_stats.md ;  python tools/pmc_traffic.py <out>/pmc
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m $1); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kernel_trace -o r -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
ls -la $OUT/kernel_trace | head
cd $R && bash tools/pmc_traffic.sh $OUT/pmc > $OUT/pmc.log 2>&1
du -sh $OUT
