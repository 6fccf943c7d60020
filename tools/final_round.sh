#!/bin/bash
# One gpurun call for a round's committed evidence, sized to come back (gpurun merges <= 64 MiB: rocpd databases are ~28 MB each, so they
# are reduced to their markdown summaries ON the box and deleted).   usage (GPU box): tools/final_round.sh <outdir under gpurun_out>
#   <out>/bench_default_kernel_stats.md   rocprofv3 --kernel-trace --stats of the default `python bench.py` (tools/rocprof_summary.py)
#   <out>/bench_default.json              its JSON line (profiled run)
#   <out>/pmc/...                         FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh) -> here: python tools/pmc_traffic.py <out>/pmc
#   <out>/bench_vda_<model>_kernel_stats.md + .json     the same for one VDA stream of each size
#   <out>/bench_plain.json                the driver's own command, WITHOUT the profiler: the compact stdout line (+ _full.json: the report)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m $1); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --full-out $OUT/bench_default_full.json > $OUT/bench_default.json 2> $OUT/bench_default.err
python $R/tools/rocprof_summary.py $OUT/kt/r_results.db "python bench.py --steps 50 --warmup 10 --no-cpu-baseline (default run: batch 1 + batched 32 + tile fit 27 + parity-class engines + sink + config 3 schemes + config 4 streams)" > $OUT/bench_default_kernel_stats.md
rm -rf $OUT/kt; tail -c 2000 $OUT/bench_default.err > $OUT/bench_default.err.tail; rm -f $OUT/bench_default.err
for cfg in "vits 336" "vitb 518"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python $R/bench.py --vda --model $1 --res $2 --steps 200 --warmup 40 --no-cpu-baseline --full-out $OUT/bench_vda_$1_full.json > $OUT/bench_vda_$1.json 2> /dev/null
  python $R/tools/rocprof_summary.py $OUT/kt/r_results.db "python bench.py --vda --model $1 --res $2 --steps 200 --warmup 40" > $OUT/bench_vda_${1}_r${2}_kernel_stats.md
  rm -rf $OUT/kt
done
cd $R && bash tools/pmc_traffic.sh $OUT/pmc > $OUT/pmc.log 2>&1
find $OUT/pmc -name "*.db" -delete 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_plain_full.json > $OUT/bench_plain.json 2> $OUT/bench_plain.err; tail -c 1500 $OUT/bench_plain.err > $OUT/bench_plain.err.tail; rm -f $OUT/bench_plain.err
du -sh $OUT
