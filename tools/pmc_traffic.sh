#!/bin/bash
# FETCH_SIZE (the counter pair of tools/profile_r5.sh's tcc1 pass: FETCH_SIZE + TCC_HIT / TCC_MISS in ONE pass exceeds the hardware's counter
# budget, rocprofv3 aborts and then hangs in its finaliser - every pass runs under `timeout`) of the FILTER launches of one bench workload under a set of params (one rocprofv3 --pmc pass each; kernel trace only).
# usage: tools/pmc_traffic.sh <name> "<bench args>" ["<bench args>" ...]
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for a in "$@"; do
  i=$((i+1)); OUT=$ROOTD/gpurun_out/traffic_${NAME}_$i; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $ROOTD/bench.py $a --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-side > $OUT/log.txt 2>&1
  python3 - $OUT "$a" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "mips_filter" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(per)
n = len(ids) // 4 if len(ids) >= 4 else len(ids)   # 4 batches (1 warm-up + 3 steps): the launches of the last batch
last = ids[-n:]
f = sum(per[i]["FETCH_SIZE"] for i in last) * 1024 * 2
print(f"[{tag}] launches/batch {n}: fetch {f / 1e9:.2f} GB per batch (x2-corrected)")
PY
done
