#!/bin/bash
# same-box A/B: rendezvous of the workgroups that share a corpus tile (param rendezvous = 1 / 0), C3 and the 1.25 M shard, + HBM traffic (PMC)
OUT=gpurun_out/r3f; mkdir -p $OUT
ROOTD=$PWD
for rep in 1 2 3; do
  for w in "c3 --steps 20 --warmup 3" "shard --rows 1250000 --steps 100 --warmup 10" "c4nq512 --rows 10000000 --nq 512 --steps 30 --warmup 3"; do
    set -- $w; name=$1; shift
    for r in 1 0; do
      python bench.py "$@" --param rendezvous=$r --no-cpu-baseline --no-side --verify-queries 16 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name rendezvous=$r rep=$rep', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms_per_step'],4), 'recall', d['verify']['recall_at_k_vs_torch_fp32'])"
    done
  done
done | tee $OUT/ab_rendezvous.txt
cd /tmp && export TMPDIR=/tmp
for r in 1 0; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $ROOTD/$OUT/pmc_rdv$r -- python3 $ROOTD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-side --param rendezvous=$r > $ROOTD/$OUT/pmc_rdv$r.log 2>&1
  f=$(find $ROOTD/$OUT/pmc_rdv$r -name "*counter_collection.csv" | head -1)
  python3 - "$f" $r <<'PY'
import csv,sys,collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    if 'filter16p' in row['Kernel_Name'] and row['Counter_Name']=='FETCH_SIZE':
        tot[row['Dispatch_Id']]+=float(row['Counter_Value'])
vals=sorted(tot.values())
# FETCH_SIZE is in 32-byte units on gfx950 per guide? report raw sum per step (last 4 launches)
import itertools
print('rendezvous=%s FETCH_SIZE per filter launch (raw units):'%sys.argv[2], [round(v) for v in list(tot.values())[-4:]], 'sum last 4', round(sum(list(tot.values())[-4:])))
PY
done | tee -a $ROOTD/$OUT/ab_rendezvous.txt
find $ROOTD/$OUT -name "*.csv" -delete 2>/dev/null
