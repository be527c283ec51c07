#!/bin/bash
# same-box A/B: the register-resident select kernel (libvodhip.so) against the round-2 one (libvodhip_old.so = the parent commit's build)
A=vod_amd/csrc/libvodhip_old.so; B=vod_amd/csrc/libvodhip.so
out=gpurun_out/ab_select.txt; mkdir -p gpurun_out; : > $out
run() { lib=$1; shift; echo -n "$(basename $lib) $*: " >> $out; VODHIP_LIB=$PWD/$lib timeout 600 python3 bench.py --no-cpu-baseline --no-side "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; v=d.get('verify') or {}
        print('ms/step %.4f  filter_ms %.4f  frac %.3f recall %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], v.get('recall_at_k_vs_torch_fp32')))
" >> $out; }
for rep in 1 2 3; do for w in "--rows 1000000 --nq 256 --steps 200 --warmup 20" "--rows 1250000 --steps 100 --warmup 10" "--rows 1250000 --steps 100 --warmup 10 --force-collective"; do run $A $w; run $B $w; done; done
for rep in 1 2; do run $A --steps 20 --warmup 3; run $B --steps 20 --warmup 3; run $A --nq 256 --steps 40 --warmup 5; run $B --nq 256 --steps 40 --warmup 5; done
cat $out
