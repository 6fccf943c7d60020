#!/bin/bash
# Kernel times of post_fused_kernel and of its PF_CUT variants (tools/build_variant.sh pfcut<n> post.hip -DPF_CUT=<n>) from rocprofv3
# kernel traces (the python loop of tools/post_bench.py is host-bound below ~25 us per call).  usage (GPU box): tools/post_cut_profile.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m $1); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in full pfcut1 pfcut2 pfcut3 pfcut4; do
  lib=$R/desktop2stereo_amd/libd2s_hip.so; [ $v != full ] && lib=$R/desktop2stereo_amd/libd2s_hip_$v.so
  [ -f $lib ] || continue
  D2S_LIB=$lib rocprofv3 --kernel-trace --stats -d $OUT/$v -o r -- python $R/tools/post_bench.py --batches 1 --no-check > $OUT/$v.log 2>&1
  python $R/tools/rocprof_summary.py $OUT/$v/r_results.db 2>/dev/null | grep -E "post_fused|percentile|shape_blur" | sed "s/^/$v: /"
done
