#!/bin/bash
# Board power / clocks / temperature while the headline bench runs (rocm-smi as an ordinary user): tools/power_probe.sh [bench args]
out=gpurun_out/power_probe.txt; mkdir -p gpurun_out; : > $out
rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>&1 | grep -v "^=\|^$" | head -20 >> $out
(python3 bench.py --no-cpu-baseline --no-verify --steps 1500 --warmup 5 "$@" > gpurun_out/power_probe_bench.json 2>/dev/null) &
pid=$!
sleep ${PROBE_DELAY:-14}   # import, index build, warm-up
for i in 1 2 3 4 5 6 7 8; do
  echo "-- sample $i" >> $out
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|temperature (sensor junction\|hotspot\|junction" | head -40 >> $out
  sleep 1.5
done
wait $pid
tail -1 gpurun_out/power_probe_bench.json | cut -c1-200 >> $out
cat $out
