#!/bin/bash
# Batch-1 frames/s of the default bench under one runtime switch at a time (same box, the baseline repeated between groups).
#   gpurun -- 'tools/env_sweep.sh "A=1" "B=0" ...'
cd ${GRAFT_REPO_ROOT:-.}
run() { echo -n "$1: "; env $1 timeout 100 python bench.py --batch 1 --also-batch 0 --steps 300 --no-cpu-baseline --sink-quality 0 --no-profile --no-parity-class --no-config3 --no-parity 2>&1 | tail -1 |
  python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))" 2>/dev/null || echo failed; }
run "D2S_BASELINE=1"
for v in "$@"; do run "$v"; done
run "D2S_BASELINE=2"
for v in "$@"; do run "$v"; done
run "D2S_BASELINE=3"
