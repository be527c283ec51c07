#!/bin/bash
# PMC passes (SQ_* and GRBM_GUI_ACTIVE, separate runs) of a bench workload for one (library, tile): MFMA-busy fraction, effective clock and
# the wave-cycle split of the FILTER launches.  usage: experiments/tools/pmc_8phase.sh <prod|abl> <tile> <name> [bench args...]
LIBSEL=$1; T=$2; NAME=$3; shift 3
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
if [ $LIBSEL = abl ]; then export VODHIP_LIB=$ROOTD/vod_amd/csrc/libvodhip_ablation.so; else unset VODHIP_LIB; fi
OUT=$ROOTD/gpurun_out/pmc8/${NAME}_${LIBSEL}_tile$T; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-side --tile $T $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -- python3 $ROOTD/bench.py $A > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python3 $ROOTD/bench.py $A > $OUT/grbm.log 2>&1
python3 - $OUT "$NAME lib=$LIBSEL tile=$T" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
def is_filter(name):  # the FILTER-mode launches of the persistent kernels (tile 8: mips_filter16p<.., 0, ..>; tiles 13 / 14: mips_filter8ph)
    return "mips_filter8ph" in name or ("mips_filter16p_kernel<" in name and ", 0, " in name.split("(")[0])
def agg(sub):
    d = {}
    for p in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if is_filter(r["Kernel_Name"]):
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return d
sq, gr = agg("sq"), agg("grbm")
dur, n = 0.0, 0
for p in glob.glob(f"{out}/grbm/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if is_filter(r["Kernel_Name"]):
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            n += 1
wc = sq.get("SQ_WAVE_CYCLES", 0.0) or 1.0
ga = gr.get("GRBM_GUI_ACTIVE", 0.0)
print(f"{tag}: {n} FILTER launches {dur * 1e3 / max(1, n):.3f} ms avg (under the counters); effective clock {ga / 8 / dur / 1e9 if dur else 0:.3f} GHz; "
      f"MFMA busy {sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (ga * 128) if ga else 0:.4f} of GPU cycles; wave cycles: waiting {sq.get('SQ_WAIT_ANY', 0) / wc:.3f}, "
      f"issue-stalled {sq.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}, issuing {sq.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}; LDS bank conflict cycles {sq.get('SQ_LDS_BANK_CONFLICT', 0):.0f}, "
      f"LDS active {sq.get('SQ_LDS_IDX_ACTIVE', 0) / wc:.3f} of wave cycles")
PY
