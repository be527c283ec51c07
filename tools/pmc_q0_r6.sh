#!/bin/bash
# Round 6: re-price the L2-miss traffic of C3 on the SHIPPED kernel (tile 14, 8-phase) for both stage orders (tile_order 0 = low-discrepancy
# permutation, 1 = row order) and with the round-4 diagnostic (kflags 128: every q-tile stages the rows of q-tile 0 - the query-tile refetches
# disappear, results wrong).  Timing from un-profiled runs; bytes / clock from separate --pmc passes (kernel-trace only).
# usage: tools/pmc_q0_r6.sh [data: iid | clustered]
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
DATA=${1:-iid}
OUT=$ROOTD/gpurun_out/r6/pmc_q0_$DATA; mkdir -p $OUT; : > $OUT/timing.txt
export VODHIP_LIB=$ROOTD/vod_amd/csrc/libvodhip_ablation.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
 for order in 0 1; do
  for kf in 0 128; do
    timeout 600 python3 $ROOTD/bench.py --data $DATA --no-side --no-cpu-baseline --no-verify --steps 10 --warmup 3 --param tile_order=$order --param kflags=$kf 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('order=$order kflags=$kf rep$rep ms', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms_per_step'],3))" | tee -a $OUT/timing.txt
  done
 done
done
for order in 0 1; do
  for kf in 0 128; do
    timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc1_o${order}_k$kf -- python3 $ROOTD/bench.py --data $DATA --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-side --param tile_order=$order --param kflags=$kf > $OUT/tcc1_o${order}_k$kf.log 2>&1
    timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc2_o${order}_k$kf -- python3 $ROOTD/bench.py --data $DATA --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-side --param tile_order=$order --param kflags=$kf > $OUT/tcc2_o${order}_k$kf.log 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for order in (0, 1):
    for kf in (0, 128):
        acc = collections.defaultdict(float); n = collections.Counter(); dur = []
        for name in ("tcc1", "tcc2"):
            for f in glob.glob(f"{out}/{name}_o{order}_k{kf}/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if "mips_filter8ph" not in row.get("Kernel_Name", ""): continue
                    acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
            if name == "tcc1":
                for f in glob.glob(f"{out}/{name}_o{order}_k{kf}/**/*kernel_trace.csv", recursive=True):
                    for row in csv.DictReader(open(f)):
                        if "mips_filter8ph" in row.get("Kernel_Name", ""): dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        launches = max(1, n["FETCH_SIZE"]); batches = launches / 3.0   # 3 FILTER launches per batch on C3
        gb = acc["FETCH_SIZE"] * 2048 / 1e9 / batches
        clk = acc["GRBM_GUI_ACTIVE"] / 8 / max(1, sum(dur)) if dur else 0
        hit = acc["TCC_HIT_sum"] / max(1.0, acc["TCC_HIT_sum"] + acc["TCC_MISS_sum"])
        print(f"tile_order {order} kflags {kf}: L2-miss bytes per batch {gb:.2f} GB = {gb / 15.36:.3f} x algorithmic   clock {clk:.3f} GHz   L2 hit {hit:.3f}   ({launches} FILTER launches profiled)")
PY
find $OUT -type f ! -name "*.txt" ! -name "*counter_collection.csv" -delete 2>/dev/null
