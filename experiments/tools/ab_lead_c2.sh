#!/bin/bash
# Is the one-query-tile regime limited by the corpus bytes in flight?  LDS-DMA lead 7 / 6 / 5 half-tiles (tiles 14 / 15 / 16, experiment build) on
# C2 and on C3 at nq 256, one lane (exclusive kernel times), interleaved.
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
export VODHIP_LIB=$ROOTD/vod_amd/csrc/libvodhip_ablation.so
for rep in 1 2; do
  for tile in 14 15 16; do
    python3 $ROOTD/bench.py --rows 1000000 --nq 256 --steps 200 --warmup 20 --tile $tile --param lanes=1 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 tile $tile rep$rep ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms_per_step'],4))"
    python3 $ROOTD/bench.py --nq 256 --steps 30 --warmup 5 --tile $tile --param lanes=1 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3nq256 tile $tile rep$rep ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms_per_step'],4))"
  done
done
