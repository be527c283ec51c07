#!/bin/bash
# Round-2 timing experiment on the diagnostic build (libvodhip_ablation.so): which part of the persistent K loop costs what.
# kflags: 1 corpus tiles always L2-hot (no HBM first touch), 2 all LDS-DMA right after the barrier, 4 static priority for
# waves 4..7, 8 no survivors (epilogue floor).  Bits 1 and 8 give wrong results by design: --no-verify.
export VODHIP_LIB=$PWD/vod_amd/csrc/libvodhip_ablation.so
out=gpurun_out/exp_knobs.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; timeout 300 python3 bench.py --no-cpu-baseline --no-verify --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('ms/step %.4f  filter_ms %.4f  qps %.0f' % (d['ms_per_step'], r['kernel_ms_per_step'], d['value']))
" >> $out; }
for f in 8 72 8 72 0 64; do run --param kflags=$f; done
for f in 8 72 8 72 24 88; do run --nq 256 --param kflags=$f; done
for f in 8 72; do run --nq 512 --param kflags=$f; done

cat $out
