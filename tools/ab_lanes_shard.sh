#!/bin/bash
OUT=gpurun_out/r5_ab_lanes_shard.txt
for rep in 1 2 3; do
  for cfg in "shard --rows 1250000 --steps 100 --warmup 10" "shard_exchange --rows 1250000 --steps 100 --warmup 10 --force-collective" "2m_nq512 --rows 2000000 --nq 512 --steps 80 --warmup 8" "c4shard --rows 5000000 --dim 1024 --dtype bf16 --nq 512 --k 200 --steps 40 --warmup 5"; do
    set -- $cfg; name=$1; shift
    for lanes in 1 2; do
      res=$(timeout 600 python bench.py "$@" --param lanes=$lanes --no-side --no-cpu-baseline --verify-queries 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['verify']; print(round(d['ms_per_step'],4), v['recall_at_k'], v['integer_twin']['ids_bit_exact'])")
      echo "$name lanes=$lanes rep$rep ms $res" | tee -a $OUT
    done
  done
done
