#!/bin/bash
# The growth / sample_div sweep on the C4 shard (5 M x 1024 bf16, 512 queries, top-200), C3 at nq 256 and the exact-f32 C2: production library, x2.
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
one() { name=$1; g=$2; sd=$3; shift 3
  python3 $ROOTD/bench.py "$@" --growth $g --param sample_div=$sd --no-side --no-cpu-baseline --verify-queries 8 2>/dev/null | tail -1 | python3 /tmp/sweep_line.py $name growth $g sample_div $sd
}
cat > /tmp/sweep_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
v = d.get("verify") or {}
print(*sys.argv[1:], "kernel_ms", round(d["roofline"]["kernel_ms_per_step"], 4), "ms", round(d["ms_per_step"], 4), "launches", d["roofline"].get("launches_per_step"), "recall", v.get("recall_at_k"))
PY
for rep in 1 2; do
  for sd in $2; do for g in $1; do one C4shard $g $sd --config c4 --rows 5000000 --steps 30 --warmup 5; done; done
done
for rep in 1 2; do
  for sd in $2; do for g in $1; do one C3nq256 $g $sd --nq 256 --steps 30 --warmup 5; done; done
done
for rep in 1 2; do
  for sd in $2; do for g in $1; do one C4shardExact $g $sd --config c4 --rows 5000000 --steps 30 --warmup 5 --exact-f32; done; done
done
