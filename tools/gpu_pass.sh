#!/bin/bash
# Round-4 GPU passes (one gpurun call each; outputs under gpurun_out/r4/<pass>/).  usage: tools/gpu_pass.sh <pass>
#   serve   GPU tests of the serving path + the boundary under the trainer's load shape (tools/bench_http_load.py, default flags)
#   serve_ab  the same load cells through round 3's Python shell (uvicorn shell, window 0 / 1 ms) for the A/B
#   tests   the whole -m gpu suite + smoke
#   bench   the default bench line
#   load / c5 / grid   see the case arms below
set -u
P=${1:-tests}
OUT=gpurun_out/r4/$P; mkdir -p $OUT
case $P in
  tests)
    python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.log 2>&1
    timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log ;;
  bench)
    timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json ;;
  serve)
    timeout 900 python -m pytest tests/test_server_gpu.py tests/test_fuzz_gpu.py tests/test_node_index_gpu.py -x -q -m gpu > $OUT/pytest_serve.log 2>&1; tail -5 $OUT/pytest_serve.log
    timeout 1500 python tools/bench_http_load.py --out $OUT/http_load_native.json > $OUT/http_load_native.log 2>&1; tail -25 $OUT/http_load_native.log | cut -c1-400 ;;
  serve_ab)
    timeout 900 python tools/bench_http_load.py --http uvicorn --micro-batch-ms 0 1 --routes fast --nq 32 64 --out $OUT/http_load_uvicorn.json > $OUT/http_load_uvicorn.log 2>&1; tail -14 $OUT/http_load_uvicorn.log | cut -c1-300 ;;
  load)   # the load cells only (no tests)
    timeout 1500 python tools/bench_http_load.py --out $OUT/http_load_native.json > $OUT/http_load_native.log 2>&1; tail -25 $OUT/http_load_native.log | cut -c1-420 ;;
  c5)     # host-time probe of the retrieval loss + the C5 side entry (with the restated CPU / op-sequence baselines)
    timeout 600 python tools/probe_h5_host.py > $OUT/probe_h5_host.json 2> $OUT/probe_h5_host.err; cat $OUT/probe_h5_host.json
    timeout 600 python tools/side_c5.py > $OUT/side_c5.json 2> $OUT/side_c5.err; tail -c 3000 $OUT/side_c5.json ;;   # (the restated baselines ride on bench.py's cpu_baseline leg)
  grid)   # the reference's published artefact on its own axes
    timeout 1200 python tools/bench_reference_grid.py --out $OUT/reference_grid.json > $OUT/reference_grid.log 2>&1; tail -15 $OUT/reference_grid.log ;;
  ab_ring)  # same-box interleaved A/B: FILTER stages of one-q-tile batches on the deep ring (tile 11, auto) vs the two-slot kernel (tile 8)
    for rep in 1 2 3; do
      for cfg in "c2 --rows 1000000 --nq 256 --steps 300 --warmup 30" "nq256 --nq 256 --steps 60 --warmup 6" "c4shard_nq256 --rows 5000000 --dim 1024 --dtype bf16 --nq 256 --k 200 --steps 60 --warmup 6"; do
        set -- $cfg; name=$1; shift
        for v in "ring" "twoslot --param ring_single_qtile=0"; do
          set -- $v; vn=$1; shift
          ms=$(timeout 600 python bench.py $(echo $cfg | cut -d" " -f2-) --no-side --no-cpu-baseline --verify-queries 16 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['verify']['recall_at_k'], d['verify']['max_abs_score_diff'])")
          echo "$name $vn rep$rep $ms" | tee -a $OUT/ab_ring.txt
        done
      done
    done ;;
  fuzz)   # randomised campaigns at HEAD: the boundary (native front: single / window / worker group / node index; uvicorn shell), search, collate
    for args in "--requests 800 --threads 16 --seed 41" "--requests 600 --threads 16 --seed 42 --wait-ms 5" "--requests 500 --threads 12 --seed 43 --group" \
                "--requests 500 --threads 12 --seed 44 --node" "--requests 400 --threads 12 --seed 45 --http uvicorn"; do
      timeout 900 python tests/fuzz/fuzz_server.py $args 2>/dev/null | tail -3 | tee -a $OUT/fuzz_server.txt
    done
    timeout 1500 python tests/fuzz/fuzz_search.py --trials 3000 --seed 404 2>/dev/null | tail -3 | tee -a $OUT/fuzz_search.txt
    timeout 1500 python tests/fuzz/fuzz_collate.py --trials 6000 --seed 405 2>/dev/null | tail -3 | tee -a $OUT/fuzz_collate.txt ;;
  openloop)  # arrivals NOT synchronised by the server: workers pause Exp(think) between requests; default policy vs never waiting for company
    timeout 900 python tools/bench_http_load.py --routes fast --clients 8 32 --nq 32 --think-ms 2 10 --out $OUT/http_openloop_default.json > $OUT/openloop_default.log 2>&1; tail -5 $OUT/openloop_default.log | cut -c1-330
    timeout 900 python tools/bench_http_load.py --routes fast --clients 8 32 --nq 32 --think-ms 2 10 --batcher-param grace_us=0 --out $OUT/http_openloop_nograce.json > $OUT/openloop_nograce.log 2>&1; tail -5 $OUT/openloop_nograce.log | cut -c1-330 ;;
  final)  # at the round's last HEAD: serving tests, the randomised campaigns on fresh seeds, the default-flag load cells
    timeout 900 python -m pytest tests/test_server_gpu.py tests/test_node_index_gpu.py -x -q -m gpu > $OUT/pytest_serve.log 2>&1; tail -3 $OUT/pytest_serve.log
    for args in "--requests 800 --threads 16 --seed 51" "--requests 500 --threads 12 --seed 53 --group" "--requests 500 --threads 12 --seed 54 --node" "--requests 400 --threads 16 --seed 56 --churn"; do
      timeout 900 python tests/fuzz/fuzz_server.py $args 2>/dev/null | tail -2 | tee -a $OUT/fuzz_server.txt
    done
    timeout 1200 python tests/fuzz/fuzz_search.py --trials 2500 --seed 504 2>/dev/null | tail -2 | tee -a $OUT/fuzz_search.txt
    timeout 900 python tests/fuzz/fuzz_collate.py --trials 4000 --seed 505 2>/dev/null | tail -2 | tee -a $OUT/fuzz_collate.txt
    timeout 1200 python tools/bench_http_load.py --routes fast --out $OUT/http_load_native.json > $OUT/http_load_native.log 2>&1; tail -12 $OUT/http_load_native.log | cut -c1-420 ;;
  campaign)  # long randomised campaigns on seeds never used before (budget that would otherwise lapse)
    timeout 2400 python tests/fuzz/fuzz_search.py --trials 9000 --seed ${SEED_A:-601} 2>/dev/null | tail -3 | tee -a $OUT/fuzz_search.txt
    timeout 1200 python tests/fuzz/fuzz_collate.py --trials 12000 --seed ${SEED_B:-602} 2>/dev/null | tail -3 | tee -a $OUT/fuzz_collate.txt
    for args in "--requests 6000 --threads 24 --seed ${SEED_C:-603} --churn" "--requests 1500 --threads 16 --seed 604 --node" "--requests 1500 --threads 16 --seed 605 --group"; do
      timeout 1200 python tests/fuzz/fuzz_server.py $args 2>/dev/null | tail -2 | tee -a $OUT/fuzz_server.txt
    done ;;
  *) echo "unknown pass $P"; exit 2 ;;
esac
