#!/bin/bash
# same-box A/B: a FILTER stage's partial last round on 128 x 128 tiles (tail_split = 1) vs on the persistent kernel (0)
OUT=gpurun_out/r3g; mkdir -p $OUT
for rep in 1 2 3; do
  for w in "c2 --rows 1000000 --nq 256 --steps 300 --warmup 30" "shard --rows 1250000 --nq 1024 --steps 100 --warmup 10" "c3 --steps 20 --warmup 3" "c3nq256 --nq 256 --steps 40 --warmup 4" "n3M_nq512 --rows 3000000 --nq 512 --steps 60 --warmup 5"; do
    set -- $w; name=$1; shift
    for r in 1 0; do
      python bench.py "$@" --param tail_split=$r --no-cpu-baseline --no-side --verify-queries 64 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name tail_split=$r rep=$rep', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms_per_step'],4), 'launches', d['roofline']['launches_per_step'], 'recall', d['verify']['recall_at_k_vs_torch_fp32'])"
    done
  done
done | tee $OUT/ab_tail_split.txt
