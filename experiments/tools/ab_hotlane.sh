#!/bin/bash
# same-box A/B: cooperative emptying of hot lanes in the filter kernel's cold path (libvodhip.so) vs the parent build (libvodhip_old.so)
A=vod_amd/csrc/libvodhip_old.so; B=vod_amd/csrc/libvodhip.so
out=gpurun_out/ab_hotlane.txt; mkdir -p gpurun_out; : > $out
run() { lib=$1; shift; echo -n "$(basename $lib) $*: " >> $out; VODHIP_LIB=$PWD/$lib timeout 600 python3 bench.py --no-cpu-baseline --no-side "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; v=d.get('verify') or {}
        print('ms/step %.4f  filter_ms %.4f  frac %.3f recall %s ids %s recov %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], v.get('recall_at_k_vs_torch_fp32'), v.get('rows_with_identical_id_order'), d['config'].get('recovery_passes')))
" >> $out; }
for rep in 1 2 3; do for w in "--steps 20 --warmup 3" "--data clustered --steps 10 --warmup 3" "--rows 1250000 --steps 100 --warmup 10" "--rows 1000000 --nq 256 --steps 200 --warmup 20"; do run $A $w; run $B $w; done; done
for w in "--rows 40000000 --dim 1024 --nq 512 --k 200 --dtype bf16 --steps 10" "--nq 256 --data clustered --steps 20"; do run $A $w; run $B $w; done
cat $out
