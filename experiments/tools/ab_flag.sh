#!/bin/bash
# same-box A/B: the exactness flag published by the final select's last workgroup (libvodhip.so) vs a 4-byte device-to-host copy
# command behind every search (libvodhip_old.so = parent build)
A=vod_amd/csrc/libvodhip_old.so; B=vod_amd/csrc/libvodhip.so
out=gpurun_out/ab_flag.txt; mkdir -p gpurun_out; : > $out
run() { lib=$1; shift; echo -n "$(basename $lib) $*: " >> $out; VODHIP_LIB=$PWD/$lib timeout 600 python3 bench.py --no-cpu-baseline --no-side "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; v=d.get('verify') or {}
        print('ms/step %.4f  filter_ms %.4f  recall %s recov %s' % (d['ms_per_step'], r['kernel_ms_per_step'], v.get('recall_at_k_vs_torch_fp32'), d['config'].get('recovery_passes')))
" >> $out; }
for rep in 1 2 3; do for w in "--rows 1000000 --nq 256 --steps 300 --warmup 20" "--rows 1250000 --steps 150 --warmup 10" "--steps 20 --warmup 3"; do run $A $w; run $B $w; done; done
run $A --data duplicates --rows 1000000 --steps 10; run $B --data duplicates --rows 1000000 --steps 10
cat $out
