#!/bin/bash
# Same-box A/B of two builds of libd2s_hip.so (boxes differ by a few % in clocks, so numbers from two gpurun calls are
# not comparable).  Put the two libraries at .ab/libA.so and .ab/libB.so (git-ignored, travels with gpurun), then:
#   gpurun -- 'BATCHES="1 8 16" tools/ab_bench.sh "<extra bench.py args>"'
cd ${GRAFT_REPO_ROOT:-.}
cp desktop2stereo_amd/libd2s_hip.so .ab/_orig.so
for rep in 1 2; do
  for v in A B; do
    cp .ab/lib$v.so desktop2stereo_amd/libd2s_hip.so
    for B in ${BATCHES:-1 16}; do
      echo -n "lib$v B=$B: "
      python bench.py --batch $B --also-batch 0 --steps 100 --no-cpu-baseline --sink-quality 0 --no-profile $1 2>&1 | tail -1 |
        python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))"
    done
  done
done
cp .ab/_orig.so desktop2stereo_amd/libd2s_hip.so
