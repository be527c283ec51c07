#!/bin/bash
# Package power and clocks while the one-query-tile loop runs (C3 store at nq 256), as shipped and with the corpus always L2-hot (diagnostic knob 1 of
# the two-slot kernel: no HBM traffic): is the 15 % that the knob gives a matter of HBM LATENCY or of the HBM traffic's share of the POWER cap?
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
export VODHIP_LIB=$ROOTD/vod_amd/csrc/libvodhip_ablation.so
for kf in 0 1; do
  out=$ROOTD/gpurun_out/r6/power_nq256_k$kf.txt; mkdir -p $ROOTD/gpurun_out/r6; : > $out
  (python3 $ROOTD/bench.py --nq 256 --tile 8 --param lanes=1 --param tile_order=1 --param kflags=$kf --no-cpu-baseline --no-verify --no-side --steps 9000 --warmup 5 > $ROOTD/gpurun_out/r6/power_bench_k$kf.json 2>/dev/null) &
  pid=$!
  sleep 14
  for i in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk" | head -6 >> $out
    sleep 1.5
  done
  wait $pid
  echo "kflags=$kf: $(tail -1 $ROOTD/gpurun_out/r6/power_bench_k$kf.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', round(d['ms_per_step'],4))")"
  grep -i "power" $out | awk '{print $NF}' | tr '\n' ' '; echo " W"
  grep -i "sclk" $out | sed 's/.*(\(.*\)Mhz).*/\1/' | tr '\n' ' '; echo " MHz (sclk level)"
done
