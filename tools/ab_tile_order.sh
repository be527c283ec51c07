#!/bin/bash
# Same-box interleaved A/B of the FILTER stages' tile order: low-discrepancy (default, tile_order=0) vs row order (tile_order=1).
# usage: tools/ab_tile_order.sh <out_file> [reps]
set -u
OUT=${1:-gpurun_out/r5_ab_tile_order.txt}; REPS=${2:-2}
for rep in $(seq 1 $REPS); do
  for cfg in "c3_iid --steps 20" "c3_clustered --data clustered --steps 10" "c3_normalized --data normalized --steps 10" "c2 --rows 1000000 --nq 256 --steps 200 --warmup 20" \
             "c2_clustered --rows 1000000 --nq 256 --steps 200 --warmup 20 --data clustered" "shard --rows 1250000 --steps 100 --warmup 10" \
             "c4shard --rows 5000000 --dim 1024 --dtype bf16 --nq 512 --k 200 --steps 40 --warmup 5" \
             "c4shard_clustered --rows 5000000 --dim 1024 --dtype bf16 --nq 512 --k 200 --steps 40 --warmup 5 --data clustered"; do
    set -- $cfg; name=$1; shift
    if [ -n "${ONLY:-}" ] && ! echo " $ONLY " | grep -q " $name "; then continue; fi
    for order in 0 1; do
      res=$(timeout 600 python bench.py "$@" --no-side --no-cpu-baseline --verify-queries 16 --param tile_order=$order 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['verify']; print(round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms_per_step'],4), 'recall', v['recall_at_k'], 'diff', v['max_abs_score_diff'], 'twin', v['integer_twin']['ids_bit_exact'], 'recov', d['config']['recovery_passes'])")
      echo "$name tile_order=$order rep$rep ms $res" | tee -a $OUT
    done
  done
done
