#!/bin/bash
# PMC counters for the GEMM micro-benchmark (one pass per counter group; no tracing flags combined).
# usage: tools/pmc_gemm.sh <outdir> [extra gemm_bench args]
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python $R/tools/gemm_bench.py $EXTRA > $OUT/$name.log 2>&1; }
EXTRA="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
