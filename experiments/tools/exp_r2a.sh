#!/bin/bash
# Round-2: first run of the bootstrap schedule + staggered persistent kernel: parity subset, then timings.
out=gpurun_out/exp_r2a.txt; mkdir -p gpurun_out; : > $out
timeout 900 python3 -m pytest tests/test_mips_gpu.py -x -q -m gpu -k "not full_size" > gpurun_out/exp_r2a_pytest.log 2>&1; echo "pytest rc=$?" >> $out; tail -5 gpurun_out/exp_r2a_pytest.log >> $out
run() { echo "== $*" >> $out; timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('ms/step %.4f  filter_ms %.4f  launches %.1f  qps %.0f  verify %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d['value'], d.get('verify')))
" >> $out; }
for t in 9 8; do
run --tile $t
run --tile $t --rows 1250000
run --tile $t --rows 1000000 --nq 256
run --tile $t --nq 256
done
run --rows 1250000 --growth 1600
run --rows 1250000 --growth 400
run --rows 1250000 --param sample_div=24
run --rows 1250000 --param sample_div=96
run --growth 1600
run --param sample_div=96
cat $out
