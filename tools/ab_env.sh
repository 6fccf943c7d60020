#!/bin/bash
# Same-box A/B of one library under two environments (engine switches read at d2s_engine_create):
#   gpurun -- 'BATCHES="1 32" tools/ab_env.sh "D2S_FUSE_PROJ=0" "D2S_FUSE_PROJ=1" ["<extra bench.py args>"]'
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for v in "$1" "$2"; do
    for B in ${BATCHES:-1 32}; do
      echo -n "$v B=$B: "
      env $v python bench.py --batch $B --also-batch 0 --steps 200 --no-cpu-baseline --sink-quality 0 --no-profile --no-parity-class --no-config3 $3 2>&1 | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'fps  depth L1/max vs ref', d.get('depth_l1_vs_ref'), d.get('depth_max_vs_ref'), 'warp lsb', d.get('warp_max_lsb'))"
    done
  done
done
