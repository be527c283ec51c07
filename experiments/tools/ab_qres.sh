#!/bin/bash
# A/B of the query-resident FILTER kernel (tile 17 / 18 (TILE=18), experiment build) against the production 8-phase kernel (tile 14), interleaved, one process each.
# usage: experiments/tools/ab_qres.sh [rounds]
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
export VODHIP_LIB=$ROOTD/vod_amd/csrc/libvodhip_ablation.so
R=${1:-3}
run() { # name, bench args...
  name=$1; shift
  for r in $(seq $R); do
    for tile in 14 ${TILE:-17}; do
      python3 $ROOTD/bench.py "$@" --tile $tile --no-side --no-cpu-baseline --verify-queries 16 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['verify']; print('$name tile $tile ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms_per_step'],4), 'recall', v['recall_at_k'], 'twin', v['integer_twin']['ids_bit_exact'], v['integer_twin']['scores_bit_exact'])"
    done
  done
}
run C2 --rows 1000000 --nq 256 --steps 200 --warmup 20
run C2_lanes1 --rows 1000000 --nq 256 --steps 200 --warmup 20 --param lanes=1
run C3nq256 --nq 256 --steps 40 --warmup 5
run C3 --steps 15 --warmup 3
run shard --rows 1250000 --steps 60 --warmup 10
