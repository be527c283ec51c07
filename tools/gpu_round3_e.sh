#!/bin/bash
set -x
OUT=gpurun_out/r3e; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_mips_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -k "not full_size" > $OUT/pytest_mips.log 2>&1; tail -n 3 $OUT/pytest_mips.log
bash tools/ab_gmax_small.sh
