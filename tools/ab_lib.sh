# same-box A/B of library builds at batch 1: tools/ab_lib.sh "<lib-suffix|default> ..."  (libd2s_hip_<suffix>.so from tools/build_variant.sh)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for L in ${LIBS:-default}; do
    [ $L = default ] && P="" || P=$PWD/desktop2stereo_amd/libd2s_hip_$L.so
    for B in ${BATCHES:-1}; do
      echo -n "lib $L B=$B: "
      D2S_LIB=$P python bench.py --batch $B --also-batch 0 --steps 200 --no-cpu-baseline --sink-quality 0 --no-profile --no-parity-class --no-config3 $1 2>&1 | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'fps  depth L1/max vs ref', d.get('depth_l1_vs_ref'), d.get('depth_max_vs_ref'))"
    done
  done
done
