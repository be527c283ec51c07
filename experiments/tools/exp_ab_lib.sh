#!/bin/bash
# A/B of two builds of the library on one box: tools/exp_ab_lib.sh <libA.so> <libB.so> [bench args...]  (alternates A B A B)
A=$1; B=$2; shift 2
out=gpurun_out/exp_ab.txt; mkdir -p gpurun_out; : > $out
run() { lib=$1; shift; echo "== $(basename $lib) $*" >> $out; VODHIP_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; v=d.get('verify') or {}
        print('ms/step %.4f  filter_ms %.4f  qps %.0f  frac %.3f recall %s' % (d['ms_per_step'], r['kernel_ms_per_step'], d['value'], r['frac'], v.get('recall_at_k_vs_torch_fp32')))
" >> $out; }
for rep in 1 2; do run $A "$@"; run $B "$@"; done
for w in "--rows 1250000" "--rows 1000000 --nq 256" "--nq 256" "--rows 40000000 --dim 1024 --nq 512 --k 200 --dtype bf16 --steps 10"; do run $A $w "$@"; run $B $w "$@"; done
cat $out
