#!/bin/bash
# float32 rows a re-scoring wave keeps in flight: 4 (production) vs 8 (libvodhip_ncr8.so), exact-f32 stores, interleaved.
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for lib in libvodhip.so libvodhip_ncr8.so; do
    export VODHIP_LIB=$ROOTD/vod_amd/csrc/$lib
    python3 $ROOTD/bench.py --rows 1000000 --nq 256 --steps 300 --warmup 30 --exact-f32 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2exact $lib rep$rep ms', round(d['ms_per_step'],4))"
    python3 $ROOTD/bench.py --rows 5000000 --dim 1024 --nq 512 --k 200 --dtype bf16 --steps 40 --warmup 5 --exact-f32 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4shard-exact-bf16 $lib rep$rep ms', round(d['ms_per_step'],4))"
    python3 $ROOTD/bench.py --rows 5000000 --dim 1024 --nq 512 --k 200 --dtype f16 --steps 40 --warmup 5 --exact-f32 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4shard-exact-f16scan $lib rep$rep ms', round(d['ms_per_step'],4))"
  done
done
python3 $ROOTD/bench.py --rows 5000000 --dim 1024 --nq 512 --k 200 --dtype bf16 --steps 40 --warmup 5 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4shard plain bf16 ms', round(d['ms_per_step'],4))"
python3 $ROOTD/bench.py --rows 5000000 --dim 1024 --nq 512 --k 200 --dtype f16 --steps 40 --warmup 5 --no-side --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4shard plain f16 ms', round(d['ms_per_step'],4))"
