#!/bin/bash
# Same-box A/B of two production libraries (experiments/_build/libvodhip_old.so = the parent commit's build, libvodhip_new.so = this tree's)
# over the bench workloads, interleaved x R, with verification.   usage (GPU box): experiments/tools/ab_libs.sh [R] [workloads...]
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
R=${1:-2}; shift || true
W=${@:-C3 shard C2 C4shard C3nq256 C3exact C3clustered C4}
cat > /tmp/ab_libs_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
v = d.get("verify") or {}
t = v.get("integer_twin") or {}
print(sys.argv[1], sys.argv[2], "ms", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms_per_step"], 4), "launches", d["roofline"].get("launches_per_step"),
      "recall", v.get("recall_at_k"), "twin", t.get("ids_bit_exact"), t.get("scores_bit_exact"))
PY
args_of() { case $1 in
  C3) echo "--steps 12 --warmup 3";; shard) echo "--rows 1250000 --steps 60 --warmup 10";; C2) echo "--rows 1000000 --nq 256 --steps 200 --warmup 20";;
  C4shard) echo "--config c4 --rows 5000000 --steps 30 --warmup 5";; C3nq256) echo "--nq 256 --steps 30 --warmup 5";; C3exact) echo "--exact-f32 --steps 12 --warmup 3";;
  C3clustered) echo "--data clustered --steps 12 --warmup 3";; C4) echo "--config c4 --steps 6 --warmup 2";; esac; }
for w in $W; do
  for rep in $(seq $R); do
    for lib in ${LIBS:-old new}; do
      VODHIP_LIB=$ROOTD/experiments/_build/libvodhip_$lib.so python3 $ROOTD/bench.py $(args_of $w) --no-side --no-cpu-baseline --verify-queries 16 2>/dev/null | tail -1 | python3 /tmp/ab_libs_line.py $w $lib
    done
  done
done
