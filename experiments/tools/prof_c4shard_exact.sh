#!/bin/bash
# Kernel breakdown of the C4 shard (5 M x 1024 bf16, 512 queries, top-200) with and without the exact-f32 store: where the mode's overhead goes.
# usage (GPU box): experiments/tools/prof_c4shard_exact.sh
set -u
cd /tmp && export TMPDIR=/tmp
ROOTD=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOTD/gpurun_out/r6; mkdir -p $OUT
for mode in plain exact; do
  X=""; [ $mode = exact ] && X="--exact-f32"
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python3 $ROOTD/bench.py --config c4 --rows 5000000 --steps 40 --warmup 10 --no-side --no-cpu-baseline --verify-queries 8 $X > $OUT/c4shard_$mode.log 2> $OUT/c4shard_$mode.err
  tail -1 $OUT/c4shard_$mode.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_per_step'])"
  f=$(find /tmp/prof_$mode -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "vodhip" in n:
        print("   %-60s calls %5s avg %10.1f us total %10.1f us" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
done
