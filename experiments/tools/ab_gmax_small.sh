#!/bin/bash
# same-box A/B: bootstrap of small stores on the 128 x 128 kernel (gmax_small = 1, default) vs on the persistent kernel (0)
OUT=gpurun_out/r3e; mkdir -p $OUT
for rep in 1 2 3; do
  for w in "c2 --rows 1000000 --nq 256 --steps 300 --warmup 30" "shard --rows 1250000 --nq 1024 --steps 150 --warmup 15" "shardfc --rows 1250000 --nq 1024 --steps 150 --warmup 15 --force-collective"; do
    set -- $w; name=$1; shift
    for g in 1 0; do
      python bench.py "$@" --param gmax_small=$g --no-cpu-baseline --no-side --verify-queries 16 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name gmax_small=$g rep=$rep', round(d['ms_per_step'],4), 'ms', round(d['roofline']['kernel_ms_per_step'],4), d['verify']['recall_at_k_vs_torch_fp32'])"
    done
  done
done | tee $OUT/ab_gmax_small.txt
