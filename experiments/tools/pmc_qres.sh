#!/bin/bash
# PMC counters of ONE FILTER launch per arm (production 8-phase kernel vs the query-resident experiment), via experiments/ubench/qres_bench.
# Separate --pmc passes (never combined with other trace domains).  usage: experiments/tools/pmc_qres.sh <nq>
set -u
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
NQ=${1:-256}
OUT=$ROOTD/gpurun_out/r6/pmc_qres_nq$NQ; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "tcc1 FETCH_SIZE GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  rm -rf $OUT/$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $ROOTD/experiments/ubench/qres_bench_BASE 1000000 $NQ 6 1e30 > $OUT/$name.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
for d in ("sq1", "tcc1"):
    for f in glob.glob(f"{out}/{d}/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            nm = "tile17_qres" if "qres" in row["Kernel_Name"] else ("tile14_8phase" if "filter8ph" in row["Kernel_Name"] else None)
            if nm:
                tot[nm][row["Counter_Name"]] += float(row["Counter_Value"]); n[nm][row["Counter_Name"]] += 1
    for f in glob.glob(f"{out}/{d}/*/*_kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            nm = "tile17_qres" if "qres" in row["Kernel_Name"] else ("tile14_8phase" if "filter8ph" in row["Kernel_Name"] else None)
            if nm: dur[nm].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for nm in sorted(tot):
    c = {k: tot[nm][k] / n[nm][k] for k in tot[nm]}
    us = sorted(dur[nm])[len(dur[nm]) // 2]
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3
    print(f"{nm}: median {us:.1f} us (profiled)  clock {clk:.2f} GHz  MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/(c.get('GRBM_GUI_ACTIVE',1)/8):.3f}  "
          f"LDS idx active / wave cycles {c.get('SQ_LDS_IDX_ACTIVE',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):.3f}  LDS_IDX_ACTIVE {c.get('SQ_LDS_IDX_ACTIVE',0):.3g}  "
          f"wait_any {c.get('SQ_WAIT_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):.3f}  wait_inst {c.get('SQ_WAIT_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):.3f}  "
          f"active_inst {c.get('SQ_ACTIVE_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):.3f}  HBM read {c.get('FETCH_SIZE',0)*2048/1e9:.3f} GB  bank conflicts {c.get('SQ_LDS_BANK_CONFLICT',0):.3g}")
PY
