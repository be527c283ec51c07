#!/bin/bash
# Round-2: stage growth sweep after the counter padding (survivors are cheaper now: do fewer, larger stages pay?)
out=gpurun_out/exp_growth.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; timeout 300 python3 bench.py --no-cpu-baseline --no-verify --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('ms/step %.4f  filter_ms %.4f  launches %.1f  qps %.0f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d['value']))
" >> $out; }
for g in 800 1200 1600 3200 8000; do run --rows 1250000 --growth $g; done
for g in 800 1600 3200 8000; do run --rows 1000000 --nq 256 --growth $g; done
for g in 800 1000 1200 1600; do run --steps 10 --growth $g; done
for sd in 48 96 192; do run --rows 1250000 --param sample_div=$sd; run --rows 1000000 --nq 256 --param sample_div=$sd; done
cat $out
