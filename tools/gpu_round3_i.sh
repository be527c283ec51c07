#!/bin/bash
# round 3: gradients wrapper, C5 with the loss kernels, randomised campaigns against the new server shell and the search
set -x
OUT=gpurun_out/r3i; mkdir -p $OUT
ROOTD=$PWD
timeout 900 python -m pytest tests/test_gradients_gpu.py tests/test_collate_device_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -n 2 $OUT/pytest.log
timeout 600 python tools/bench_c5.py > $OUT/c5_latency.json 2>$OUT/c5.err; cat $OUT/c5_latency.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/$OUT/c5_full_prof -- python3 $ROOTD/tools/bench_c5.py > $ROOTD/$OUT/c5_full_prof.log 2>&1
cd $ROOTD
find $OUT/c5_full_prof -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
for seed in 1 2; do
  timeout 900 python tests/fuzz/fuzz_server.py --requests 600 --threads 8 --seed $seed 2>&1 | tail -1
  timeout 900 python tests/fuzz/fuzz_server.py --requests 400 --threads 8 --seed $((seed+10)) --wait-ms 0 2>&1 | tail -1
done | tee $OUT/fuzz_server.txt
timeout 900 python tests/fuzz/fuzz_server.py --requests 400 --threads 8 --seed 7 --group 2>&1 | tail -1 | tee -a $OUT/fuzz_server.txt
timeout 1200 python tests/fuzz/fuzz_search.py --trials 3000 --seconds 900 --seed 31 2>&1 | tail -2 | tee $OUT/fuzz_search.txt
