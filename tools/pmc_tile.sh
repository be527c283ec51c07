#!/bin/bash
# PMC passes (SQ_* and GRBM_GUI_ACTIVE) of the headline bench for one kernel variant: tools/pmc_tile.sh <tile>
# prints MFMA-busy fraction and effective clock of the FILTER launches.
export VODHIP_LIB=${VODHIP_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/vod_amd/csrc/libvodhip_ablation.so}
T=$1; ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOTD/gpurun_out/pmc_tile$T; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-verify --param tile=$T"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $ROOTD/bench.py $A > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python3 $ROOTD/bench.py $A > $OUT/grbm.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
def rows(sub):
    r = []
    for p in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        r += list(csv.DictReader(open(p)))
    return r
def agg(sub):
    d = {}
    for r in rows(sub):
        if "mips_filter16" not in r["Kernel_Name"] or "Li2E" in r["Kernel_Name"] and False:
            continue
        d.setdefault(r["Counter_Name"], 0.0)
        d[r["Counter_Name"]] += float(r["Counter_Value"])
    return d
sq, gr = agg("sq"), agg("grbm")
dur = 0.0
for p in glob.glob(f"{out}/grbm/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "mips_filter16" in r["Kernel_Name"]:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
print("filter kernels: total %.3f s of launches" % dur)
if gr.get("GRBM_GUI_ACTIVE") and dur:
    print("effective clock %.3f GHz" % (gr["GRBM_GUI_ACTIVE"] / 8 / dur / 1e9))
if sq.get("SQ_BUSY_CYCLES"):
    # MFMA busy is per SIMD-cycle summed; normalise like summarize_profiles does: / (BUSY_CYCLES/ (n_xcd?)) - print raw ratios
    print({k: v for k, v in sq.items()})
    wc = sq["SQ_WAVE_CYCLES"]
    print("wait_any %.3f wait_inst %.3f active %.3f of wave cycles" % (sq["SQ_WAIT_ANY"] / wc, sq["SQ_WAIT_INST_ANY"] / wc, sq["SQ_ACTIVE_INST_ANY"] / wc))
PY
