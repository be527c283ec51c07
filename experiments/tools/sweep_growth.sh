#!/bin/bash
# Stage growth factor (x100) and bootstrap sample divisor on C3 / the 1.25 M-row shard / C2, production library, interleaved x2.
# usage (GPU box): experiments/tools/sweep_growth.sh "400 500 600 800" "96"
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
cat > /tmp/sweep_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
v = d.get("verify") or {}
print(*sys.argv[1:], "kernel_ms", round(d["roofline"]["kernel_ms_per_step"], 4), "ms", round(d["ms_per_step"], 4), "launches", d["roofline"].get("launches_per_step"), "recall", v.get("recall_at_k"))
PY
one() { name=$1; g=$2; sd=$3; shift 3
  python3 $ROOTD/bench.py "$@" --growth $g --param sample_div=$sd --no-side --no-cpu-baseline --verify-queries 8 2>/dev/null | tail -1 | python3 /tmp/sweep_line.py $name growth $g sample_div $sd
}
for rep in 1 2; do
  for sd in $2; do for g in $1; do one C3 $g $sd --steps 12 --warmup 3; done; done
done
for rep in 1 2; do
  for sd in $2; do for g in $1; do one shard $g $sd --rows 1250000 --steps 60 --warmup 10; done; done
done
for rep in 1 2; do
  for sd in $2; do for g in $1; do one C2 $g $sd --rows 1000000 --nq 256 --steps 200 --warmup 20; done; done
done
