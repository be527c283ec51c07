cd /tmp && export TMPDIR=/tmp
export VODHIP_LIB=$GRAFT_REPO_ROOT/vod_amd/csrc/libvodhip_ablation.so
tl() { # timeline of the last batch
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$1 -- python3 $GRAFT_REPO_ROOT/bench.py ${@:2} --steps 4 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
python3 - $1 <<'PY'
import csv,glob,os,sys
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+f"/gpurun_out/tl_{sys.argv[1]}/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
idxs=[i for i,r in enumerate(rows) if "mips_prepare" in r["Kernel_Name"]]
a=idxs[-2]
t0=int(rows[a]["Start_Timestamp"])
print("==",sys.argv[1])
for r in rows[a:idxs[-1]]:
    print("%8.1f %8.1f %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:60]))
PY
}
tl c2 --rows 1000000 --nq 256
tl c2_nosurv --rows 1000000 --nq 256 --param kflags=8
tl c2_g4 --rows 1000000 --nq 256 --growth 400
tl c2_g16 --rows 1000000 --nq 256 --growth 1600
tl shard --rows 1250000
tl shard_nosurv --rows 1250000 --param kflags=8
tl shardfc --rows 1250000 --force-collective
