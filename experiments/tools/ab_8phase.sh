#!/bin/bash
# Same-box interleaved A/B, experiment build (libvodhip_ablation.so): tile 8 (production K loop) vs tile 13 (guide's 8-phase K loop).
# usage: experiments/tools/ab_8phase.sh <out_file> [reps]
set -u
OUT=${1:-gpurun_out/r5_ab_8phase.txt}; REPS=${2:-3}
ABL=$PWD/vod_amd/csrc/libvodhip_ablation.so
for rep in $(seq 1 $REPS); do
  for cfg in "c3 --steps 20" "shard --rows 1250000 --steps 100 --warmup 10" "c4shard --rows 5000000 --dim 1024 --dtype bf16 --nq 512 --k 200 --steps 40 --warmup 5" "c2 --rows 1000000 --nq 256 --steps 200 --warmup 20"; do
    set -- $cfg; name=$1; shift
    if [ -n "${ONLY:-}" ] && ! echo " $ONLY " | grep -q " $name "; then continue; fi
    IFS=';' read -ra ARMLIST <<< "${ARMS:-prod 8 0;abl 8 0;abl 13 0}"
    for arm in "${ARMLIST[@]}"; do
      set -- $arm; lib=$1; tile=$2; kf=${3:-0}
      if [ $lib = abl ]; then export VODHIP_LIB=$ABL; else unset VODHIP_LIB; fi
      set -- $(echo $cfg | cut -d" " -f2-)
      res=$(timeout 600 python bench.py "$@" --tile $tile --param kflags=$kf --no-side --no-cpu-baseline --verify-queries 16 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['verify']; r=d['roofline']; print(round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms_per_step'],4), 'mfma_frac', round(r['mfma_frac_of_2.5PF'],4), 'recall', v['recall_at_k'], 'diff', v['max_abs_score_diff'], 'twin', v['ids_bit_exact_on_integer_twin']['ids_bit_exact'])")
      echo "$name lib=$lib tile=$tile kflags=$kf rep$rep ms $res" | tee -a $OUT
    done
  done
done
