#!/bin/bash
# Bench lines only (no profiler), one line per workload: tools/quick_bench.sh [extra bench args applied to every workload]
out=gpurun_out/quick_bench.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; timeout 600 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; v=d.get('verify') or {}
        print('ms/step %.4f  filter_ms %.4f  launches %.1f  qps %.0f  frac %.3f recovery %s recall %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d['value'], r['frac'], d['config'].get('recovery_passes'), v.get('recall_at_k_vs_torch_fp32')))
" >> $out; }
run "$@"
run --rows 1250000 "$@"
run --rows 1250000 --force-collective "$@"
run --rows 1000000 --nq 256 "$@"
run --nq 256 "$@"
run --data clustered "$@"
run --rows 40000000 --dim 1024 --nq 512 --k 200 --dtype bf16 --steps 10 "$@"
run --nq 64 "$@"
run --nq 32 --k 10 "$@"
cat $out
