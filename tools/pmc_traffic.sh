#!/bin/bash
# HBM-side traffic of the bench step per kernel class (counters only: one rocprofv3 --pmc pass per counter, no tracing flags).
# usage (GPU box): tools/pmc_traffic.sh <outdir>      then here: python tools/pmc_traffic.py <outdir>
set -u
OUT=$(realpath -m $1); mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(pwd)}
# identity of the code these passes ran on (bench.py refuses the figures for any other tree) + when
(cd $R && python -m desktop2stereo_amd.build --digest) > $OUT/kernel_sources.sha256
date -u +%Y-%m-%dT%H:%M:%SZ > $OUT/date_utc.txt
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 $R/tools/ubench/copy_calib.hip -o /tmp/copy_calib || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/calib/$c -o calib --output-format csv -- /tmp/copy_calib > $OUT/calib_$c.log 2>&1
  for B in 1 32; do
    rocprofv3 --pmc $c -d $OUT/bench_b$B/$c -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --batch $B --also-batch 0 \
      --no-profile --no-cpu-baseline --no-parity-class --no-parity --no-config3 --sink-quality 0 --ingest own > $OUT/bench_b${B}_$c.log 2>&1
  done
done
ls $OUT
