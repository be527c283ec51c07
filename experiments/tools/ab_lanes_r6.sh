#!/bin/bash
# One lane vs two lanes (param "lanes") on batches above 256 queries, production library, interleaved x3 (bench.py runs one search ahead).
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
one() { name=$1; l=$2; shift 2
  python3 $ROOTD/bench.py "$@" --no-side --no-cpu-baseline --verify-queries 8 --param lanes=$l 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); v=d.get('verify') or {}; print('$name', 'lanes', $l, 'ms', round(d['ms_per_step'],4), 'recall', v.get('recall_at_k'), (v.get('integer_twin') or {}).get('ids_bit_exact'))"
}
for rep in 1 2 3; do
  for l in 1 2; do one shard $l --rows 1250000 --steps 60 --warmup 10; done
  for l in 1 2; do one shardfc $l --rows 1250000 --steps 60 --warmup 10 --force-collective; done
  for l in 1 2; do one C3 $l --steps 12 --warmup 3; done
  for l in 1 2; do one C4shard $l --config c4 --rows 5000000 --steps 30 --warmup 5; done
  for l in 1 2; do one C3exact $l --exact-f32 --steps 12 --warmup 3; done
done
