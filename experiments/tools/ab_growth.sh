#!/bin/bash
# same-box A/B of the planner rule "a short first stage + ONE stage with the rest -> three stages at growth 4" (libvodhip.so) against the
# parent build (libvodhip_old.so), on the workload it changes: the 1.25 M-row shard of the headline (without / with the exchange step)
A=vod_amd/csrc/libvodhip_old.so; B=vod_amd/csrc/libvodhip.so
out=gpurun_out/ab_growth.txt; mkdir -p gpurun_out; : > $out
run() { lib=$1; shift; echo -n "$(basename $lib) $*: " >> $out; VODHIP_LIB=$PWD/$lib timeout 600 python3 bench.py --no-cpu-baseline --no-side "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; v=d.get('verify') or {}
        print('ms/step %.4f  filter_ms %.4f  launches %s recall %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], v.get('recall_at_k_vs_torch_fp32')))
" >> $out; }
for rep in 1 2 3 4; do for w in "--rows 1250000 --steps 150 --warmup 10" "--rows 1250000 --steps 150 --warmup 10 --force-collective"; do run $A $w; run $B $w; done; done
cat $out
