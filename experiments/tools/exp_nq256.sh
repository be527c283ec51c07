#!/bin/bash
# Round-2 experiment: which tile family is fastest at nq = 256 (HBM-bound regime), 10 M and 1 M rows.
out=gpurun_out/exp_nq256.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; python3 bench.py --no-cpu-baseline --no-verify --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('ms/step %.4f  filter_ms %.4f  launches %.1f  qps %.0f hbm_frac %.3f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d['value'], r['hbm_frac_at_8TBps']))
" >> $out; }
for t in 10 46 9 1 42; do run --nq 256 --tile $t; done
for t in 10 46 9; do run --nq 256 --rows 1000000 --tile $t; done
for t in 46 10; do run --nq 128 --tile $t; done
for t in 9 10; do run --nq 512 --tile $t; done
cat $out
