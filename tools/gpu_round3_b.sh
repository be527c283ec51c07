#!/bin/bash
# round 3, collate pass: parity of the rewritten merge / sampling / flatten kernels + C5 latency + rocprof kernel trace of C5
set -x
OUT=gpurun_out/r3b; mkdir -p $OUT
ROOTD=$(pwd)
timeout 1800 python -m pytest tests/test_hybrid_gpu.py tests/test_sampling_gpu.py tests/test_collate_gpu.py tests/test_collate_device_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $OUT/pytest_collate.log 2>&1
tail -n 15 $OUT/pytest_collate.log
timeout 600 python tools/bench_c5.py > $OUT/c5_latency.json 2> $OUT/c5_latency.err
cat $OUT/c5_latency.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/$OUT/c5_prof -- python3 $ROOTD/tools/bench_c5.py --collate-only > $ROOTD/$OUT/c5_prof.log 2>&1
cd $ROOTD
find $OUT/c5_prof -type f ! -name "*kernel_stats.csv" ! -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT/c5_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -n 12
VODHIP_LIB=$PWD/vod_amd/csrc/libvodhip_ablation.so python tools/probe_c5.py > $OUT/c5_probe.json 2>$OUT/c5_probe.err; cat $OUT/c5_probe.json
