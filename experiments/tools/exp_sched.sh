#!/bin/bash
# Round-2 experiment: cost of survivor-dense epilogues vs number of stages (old geometric schedule, knobs only).
out=gpurun_out/exp_sched.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; python3 bench.py --no-cpu-baseline --no-verify --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('ms/step %.4f  filter_ms %.4f  launches %.1f  qps %.0f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d['value']))
" >> $out; }
S="--rows 1250000"
run $S
run $S --param cand_cap=16384 --growth 1600
run $S --param cand_cap=16384 --growth 3200
run $S --param cand_cap=16384 --growth 1600 --param dense_rows=4096
run $S --param cand_cap=16384 --growth 6400 --param dense_rows=4096
run $S --param cand_cap=16384 --growth 1800 --param dense_rows=4096
run $S --param cand_cap=8192 --growth 800 --param dense_rows=8192
C2="--rows 1000000 --nq 256"
run $C2
run $C2 --param cand_cap=16384 --growth 1600 --param dense_rows=4096
run $C2 --param cand_cap=16384 --growth 3200
run $C2 --param cand_cap=16384 --growth 6400 --param dense_rows=4096
run --steps 10
run --steps 10 --param cand_cap=16384 --growth 1600 --param dense_rows=4096
run --steps 10 --param cand_cap=16384 --growth 5000 --param dense_rows=4096
cat $out
