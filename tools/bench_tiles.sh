#!/bin/bash
# A/B the filter-kernel variants on the bench workload; prints q/s, ms/step, kernel TFLOP/s, kernel ms/step
# usage: bench_tiles.sh "<extra bench args>" tile...
EXTRA="$1"; shift
for t in "$@"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --tile $t $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tile $t $EXTRA', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['kernel_ms_per_step'],3))"
done
