#!/bin/bash
# round 3, boundary pass: server tests on the GPU engine, single-client latency, concurrent-client load
set -x
OUT=gpurun_out/r3c; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_server_gpu.py -x -q -m gpu > $OUT/pytest_server.log 2>&1; tail -n 3 $OUT/pytest_server.log
timeout 600 python tools/bench_http.py 1000000 768 > $OUT/http_latency.json 2> $OUT/http_latency.err; cat $OUT/http_latency.json
timeout 1500 python tools/bench_http_load.py --out gpurun_out/r3c/http_load.json > $OUT/http_load.log 2> $OUT/http_load.err; tail -n 40 $OUT/http_load.log | cut -c1-330
