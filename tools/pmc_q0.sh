#!/bin/bash
# Round-4 experiment on the 1.36x L2-miss traffic of C3: is it the QUERY tiles that are re-fetched?  Diagnostic build,
# knob 128 = every q-tile reads the rows of q-tile 0 (an XCD's query working set shrinks from 1024 to 256 rows; results wrong).
# FETCH_SIZE / TCC_MISS of the filter launches with and without the knob (separate --pmc passes, kernel-trace only), plus timing.
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out/r4/pmc_q0; mkdir -p $OUT
export VODHIP_LIB=$ROOTD/vod_amd/csrc/libvodhip_ablation.so
cd /tmp && export TMPDIR=/tmp
for kf in 0 128; do
  for rep in 1 2; do
    timeout 600 python3 $ROOTD/bench.py --no-side --no-cpu-baseline --no-verify --steps 10 --warmup 3 --param kflags=$kf 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kflags=$kf rep$rep ms', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms_per_step'],3))" | tee -a $OUT/timing.txt
  done
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc1_k$kf -- python3 $ROOTD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-side --param kflags=$kf > $OUT/tcc1_k$kf.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc2_k$kf -- python3 $ROOTD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-side --param kflags=$kf > $OUT/tcc2_k$kf.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for kf in (0, 128):
    for name in ("tcc1", "tcc2"):
        for f in glob.glob("$OUT/%s_k%d/**/*counter_collection.csv" % (name, kf), recursive=True):
            acc = collections.defaultdict(float); n = collections.Counter()
            for row in csv.DictReader(open(f)):
                if "mips_filter" not in row.get("Kernel_Name", ""): continue
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
            print("kflags", kf, name, {k: (v, n[k]) for k, v in acc.items()})
PY
find $OUT -type f ! -name "*.txt" ! -name "*counter_collection.csv" -delete 2>/dev/null
