#!/bin/bash
# round 3, first GPU pass: the new parity tests (C2 at full size, bench plumbing, group-server robustness) + the default bench line
set -x
mkdir -p gpurun_out/r3a
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r3a/smoke.log 2>&1
timeout 1500 python -m pytest tests/test_mips_gpu.py -x -q -m gpu -k "c2 or C2" > gpurun_out/r3a/pytest_c2.log 2>&1
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_server_gpu.py -x -q -m gpu > gpurun_out/r3a/pytest_bench_server.log 2>&1
timeout 900 python bench.py > gpurun_out/r3a/bench_default.json 2> gpurun_out/r3a/bench_default.err
tail -3 gpurun_out/r3a/*.log
cat gpurun_out/r3a/bench_default.json | tail -1 | head -c 6000
