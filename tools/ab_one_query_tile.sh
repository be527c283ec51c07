#!/bin/bash
OUT=gpurun_out/r5_ab_1qtile.txt
for rep in 1 2 3; do
  for cfg in "c2 --rows 1000000 --nq 256 --steps 300 --warmup 30" "c3_nq256 --nq 256 --steps 60 --warmup 6" "c4shard_nq256 --rows 5000000 --dim 1024 --dtype bf16 --nq 256 --k 200 --steps 60 --warmup 6"; do
    set -- $cfg; name=$1; shift
    for tile in 8 14; do
      for lanes in 1 2; do
        res=$(timeout 600 python bench.py "$@" --tile $tile --param lanes=$lanes --no-side --no-cpu-baseline --verify-queries 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['verify']['recall_at_k'])")
        echo "$name tile=$tile lanes=$lanes rep$rep ms $res" | tee -a $OUT
      done
    done
  done
done
