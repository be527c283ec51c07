#!/bin/bash
# round 3: do the experiment FILTER kernels (deep ring 10 / 11, 384 x 256 tile 12) help in the single-q-tile regime (C2, nq = 256 on 10 M)?
export VODHIP_LIB=$PWD/vod_amd/csrc/libvodhip_ablation.so
for rep in 1 2; do
for w in "c2 --rows 1000000 --nq 256 --steps 300 --warmup 30" "c3nq256 --rows 10000000 --nq 256 --steps 40 --warmup 4" "shard --rows 1250000 --nq 1024 --steps 100 --warmup 10"; do
  set -- $w; name=$1; shift
  for t in 8 9 10 11 12; do
    python bench.py "$@" --tile $t --no-cpu-baseline --no-side --verify-queries 16 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name tile=$t rep=$rep', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms_per_step'],4), 'recall', d['verify']['recall_at_k_vs_torch_fp32'])"
  done
done
done
