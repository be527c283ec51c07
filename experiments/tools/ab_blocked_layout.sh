#!/bin/bash
# Is the one-query-tile regime sensitive to DRAM locality?  Diagnostic knob 64 of the two-slot kernel (tile 8, ablation build): the corpus is read AS IF
# stored [tile][k-slice][256 rows][128 B] (one K-slice of a tile = 32 contiguous KB instead of 256 x 128 B at stride 1536; results wrong, timing only).
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
export VODHIP_LIB=$ROOTD/vod_amd/csrc/libvodhip_ablation.so
for rep in 1 2; do
  for kf in 0 64; do
    python3 $ROOTD/bench.py --rows 1000000 --nq 256 --steps 200 --warmup 20 --tile 8 --param lanes=1 --param kflags=$kf --param tile_order=1 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 tile 8 kflags $kf rep$rep kernel_ms', round(d['roofline']['kernel_ms_per_step'],4))"
    python3 $ROOTD/bench.py --nq 256 --steps 30 --warmup 5 --tile 8 --param lanes=1 --param kflags=$kf --param tile_order=1 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3nq256 tile 8 kflags $kf rep$rep kernel_ms', round(d['roofline']['kernel_ms_per_step'],4))"
    python3 $ROOTD/bench.py --nq 64 --steps 30 --warmup 5 --param lanes=1 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3nq64 (ring kernel, reference) rep$rep kernel_ms', round(d['roofline']['kernel_ms_per_step'],4), 'GB/s', round(d['roofline']['achieved']))"
  done
done
