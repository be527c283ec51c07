#!/bin/bash
# Round-6 measurement pass on the GPU box: for every workload a bench line, a rocprofv3 kernel trace + stats of the same
# command, and three separate --pmc passes (never combined with other trace domains).  Results land in gpurun_out/r06/.
# usage: tools/profile_r6.sh [workload ...]   (default: all)
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out/r06; mkdir -p $OUT
declare -A W
W[c3]=""
W[shard]="--rows 1250000"
W[c2]="--rows 1000000 --nq 256"
W[c4]="--rows 40000000 --dim 1024 --nq 512 --k 200 --dtype bf16"
W[c3nq256]="--nq 256"
W[c3clustered]="--data clustered"
W[shardfc]="--rows 1250000 --force-collective"
W[c4shard]="--rows 5000000 --dim 1024 --nq 512 --k 200 --dtype bf16"
W[c3exact]="--exact-f32"
W[c2exact]="--rows 1000000 --nq 256 --exact-f32"
LIST="${@:-c3 shard c2 c4 c4shard c3nq256 c3clustered shardfc c3exact c2exact}"
cd /tmp && export TMPDIR=/tmp
for w in $LIST; do
  a="${W[$w]}"
  extra="--no-cpu-baseline --no-side"; [ "$w" = "c3" ] && extra=""   # c3 = the default command: headline + side workloads + CPU baseline
  steps=""; case $w in c2|c2exact) steps="--steps 200 --warmup 20";; shard|shardfc|c4shard) steps="--steps 60 --warmup 10";; esac   # (a 20-step C2 run is 9 ms on a cold device: round 5's r05_c2_bench.json read 0.567 ms for that reason)
  timeout 900 python3 $ROOTD/bench.py $a $extra $steps > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err
  tail -1 $OUT/${w}_bench.json | cut -c1-400
  P="--steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-side"
  # one-query-tile workloads run on two lanes (two searches overlap on the device): the kernel trace of those is taken with ONE lane, so that
  # every kernel's duration and the last batch's timeline are exclusive (the bench line above them is the two-lane figure)
  case $w in c2|c2exact|c3nq256) P="$P --param lanes=1";; esac
  rm -rf $OUT/${w}_prof $OUT/${w}_pmc_sq1 $OUT/${w}_pmc_tcc1 $OUT/${w}_pmc_tcc2   # (a re-run must not leave two traces side by side)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_prof -- python3 $ROOTD/bench.py $a $P > $OUT/${w}_prof.log 2>&1
  [ "$w" = "shardfc" ] && continue   # (the exchange step: kernel trace only - what the RCCL all-gather and the merge cost per step)
  for pass in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "tcc1 FETCH_SIZE GRBM_GUI_ACTIVE" "tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    set -- $pass; name=$1; shift
    timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${w}_pmc_$name -- python3 $ROOTD/bench.py $a --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-side > $OUT/${w}_pmc_$name.log 2>&1
  done
  # keep the merged-back payload small: stats + counter CSVs only
  find $OUT/${w}_prof $OUT/${w}_pmc_* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete 2>/dev/null
  find $OUT/${w}_pmc_* -name "*kernel_trace.csv" -delete 2>/dev/null
done
# C5: every kernel of the collate-side chain and of the retrieval loss (what the C5 side entry's launch counts refer to)
if [[ " $LIST " == *" c3 "* ]]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_prof -- python3 $ROOTD/tools/bench_c5.py > $OUT/c5_latency.json 2> $OUT/c5_prof.log
  find $OUT/c5_prof -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
fi
ls $OUT | head -80
