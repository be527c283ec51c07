#!/bin/bash
# round 3, ingest + group pass
set -x
OUT=gpurun_out/r3d; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_server_gpu.py tests/test_mips_gpu.py -x -q -m gpu -k "eight_workers or zarr or save_load or add_from or incremental or roundtrip_config1" > $OUT/pytest.log 2>&1; tail -n 3 $OUT/pytest.log
free -g | head -2
timeout 1500 python tools/bench_ingest.py --out gpurun_out/r3d/ingest.json > $OUT/ingest.log 2> $OUT/ingest.err; cat $OUT/ingest.log; tail -3 $OUT/ingest.err
