#!/bin/bash
# Collect PMC counters for the bench workload in separate passes (never combined with trace domains other than kernel-trace).
# usage: tools/pmc.sh <out_subdir> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py $BENCH_ARGS > $OUT/$name.log 2>&1; }
BENCH_ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline --no-verify}"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls -R $OUT | head -30
