#!/bin/bash
# Where the K-split FILTER kernel (tile 18) loses its time: the C3 and C2 batches with one part of the step removed at a time
# (libraries from experiments/tools/build_ksplit_parts.sh; results are wrong with a part removed - bytes / timing only), tile 14 beside them.
# usage (GPU box): experiments/tools/ab_ksplit_parts.sh "0 1 2 4 8 16 31"
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
one() { # lib, tile, name, bench args
  lib=$1; tile=$2; name=$3; shift 3
  VODHIP_LIB=$lib python3 $ROOTD/bench.py "$@" --tile $tile --no-side --no-cpu-baseline --verify-queries 0 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '$(basename $lib)', 'tile $tile kernel_ms', round(d['roofline']['kernel_ms_per_step'],4))"
}
for rep in 1 2; do
  one $ROOTD/vod_amd/csrc/libvodhip_ablation.so 14 C3 --steps 12 --warmup 3
  for b in $1; do one $ROOTD/experiments/_build/libvodhip_ks$b.so 18 C3 --steps 12 --warmup 3; done
done
one $ROOTD/vod_amd/csrc/libvodhip_ablation.so 14 C2 --rows 1000000 --nq 256 --steps 200 --warmup 20
for b in $1; do one $ROOTD/experiments/_build/libvodhip_ks$b.so 18 C2 --rows 1000000 --nq 256 --steps 200 --warmup 20; done
