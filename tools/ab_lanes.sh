#!/bin/bash
# Same-box interleaved A/B of the two-lane pipeline ("lanes" = 1: every search on the caller's stream; 2: consecutive searches alternate
# between two workspaces / streams and overlap on the device).  usage: tools/ab_lanes.sh <out_file> [reps]
set -u
OUT=${1:-gpurun_out/r5_ab_lanes.txt}; REPS=${2:-3}
for rep in $(seq 1 $REPS); do
  for cfg in "c2 --rows 1000000 --nq 256 --steps 300 --warmup 30" "c2_exact --rows 1000000 --nq 256 --steps 300 --warmup 30 --exact-f32" \
             "c3_nq256 --nq 256 --steps 60 --warmup 6" "1m_nq64 --rows 1000000 --nq 64 --steps 400 --warmup 40" "1m_nq32_k10 --rows 1000000 --nq 32 --k 10 --steps 400 --warmup 40" \
             "shard --rows 1250000 --steps 100 --warmup 10" "c4shard --rows 5000000 --dim 1024 --dtype bf16 --nq 512 --k 200 --steps 40 --warmup 5" "c3 --steps 20"; do
    set -- $cfg; name=$1; shift
    if [ -n "${ONLY:-}" ] && ! echo " $ONLY " | grep -q " $name "; then continue; fi
    for lanes in 1 2; do
      res=$(timeout 600 python bench.py "$@" --no-side --no-cpu-baseline --verify-queries 16 --param lanes=$lanes 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['verify']; print(round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms_per_step'],4), 'recall', v['recall_at_k'], 'diff', v['max_abs_score_diff'], 'twin', v['integer_twin']['ids_bit_exact'])")
      echo "$name lanes=$lanes rep$rep ms $res" | tee -a $OUT
    done
  done
done
