#!/bin/bash
# Timing (and, with VQ=16, verification) of the diagnostic 8-phase libraries (experiments/tools/build_p8_parts.sh) on C3 / C2 / the 1.25 M-row
# shard, interleaved.   usage (GPU box): [VQ=16] experiments/tools/ab_p8_parts.sh "base noepi ..." [rounds]
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
R=${2:-2}
cat > /tmp/ab_p8_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
v = d.get("verify") or {}
t = v.get("integer_twin") or {}
print(sys.argv[1], sys.argv[2], "kernel_ms", round(d["roofline"]["kernel_ms_per_step"], 4), "ms", round(d["ms_per_step"], 4),
      "recall", v.get("recall_at_k"), "twin", t.get("ids_bit_exact"), t.get("scores_bit_exact"))
PY
one() { # tag, name, bench args
  tag=$1; name=$2; shift 2
  VODHIP_LIB=$ROOTD/experiments/_build/libvodhip_p8_$tag.so python3 $ROOTD/bench.py "$@" --tile 14 --no-side --no-cpu-baseline --verify-queries ${VQ:-0} 2>/dev/null | tail -1 | python3 /tmp/ab_p8_line.py $name $tag
}
for rep in $(seq $R); do
  for t in $1; do one $t C3 --steps 12 --warmup 3; done
done
for rep in $(seq $R); do
  for t in $1; do one $t C2 --rows 1000000 --nq 256 --steps 200 --warmup 20 --param lanes=1; done
done
for rep in $(seq $R); do
  for t in $1; do one $t shard --rows 1250000 --steps 60 --warmup 10; done
done
