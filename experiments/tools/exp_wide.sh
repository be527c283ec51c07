#!/bin/bash
# 384 x 256 tile kernel (tile 12, experiment builds: make -C vod_amd/csrc ABLATION=1 EXPERIMENTS=1) vs the 256 x 256 persistent kernel
# (tile 8): parity (PARITY=1), then bench lines on one box.
EXP_LIB=${EXP_LIB:-$PWD/vod_amd/csrc/libvodhip_ablation.so}   # tiles 10..12 live in the experiment build; every other tile runs on the product
out=gpurun_out/exp_wide.txt; mkdir -p gpurun_out; : > $out
if [ -n "$PARITY" ]; then
  sed -e "s/^TILES = \[1, 8, 9, 42, 46\]/TILES = [1, 8, 12, 42, 46]/" -e 's/parametrize("tile", \[9, 8\])/parametrize("tile", [12, 8])/' -e 's/parametrize("tile", \[0, 1, 8, 9, 42, 46\])/parametrize("tile", [0, 1, 8, 12, 42, 46])/' -e 's/parametrize("tile", \[0, 8, 9\])/parametrize("tile", [0, 8, 12])/' tests/test_mips_gpu.py > tests/test_mips_wide_tmp_gpu.py
  VODHIP_LIB=$EXP_LIB timeout 900 python -m pytest tests/test_mips_wide_tmp_gpu.py -x -q -m gpu -k "not full_size" 2>&1 | tail -5 >> $out; rm -f tests/test_mips_wide_tmp_gpu.py
fi
run() { echo "== $*" >> $out; lib=$PWD/vod_amd/csrc/libvodhip.so; case "$*" in *tile=1[0-2]*) lib=$EXP_LIB;; esac; VODHIP_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; v=d.get('verify') or {}
        print('ms/step %.4f  filter_ms %.4f  launches %.1f  qps %.0f  recovery %s recall %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d['value'], d['config'].get('recovery_passes'), v.get('recall_at_k_vs_torch_fp32')))
" >> $out; }
TILES=${TILES:-"8 12"}
for t in $TILES $TILES; do run --param tile=$t; done
for t in $TILES; do
run --param tile=$t --rows 1250000
run --param tile=$t --rows 1000000 --nq 256
run --param tile=$t --nq 256
run --param tile=$t --rows 40000000 --dim 1024 --nq 512 --k 200 --dtype bf16 --steps 10
done
cat $out
