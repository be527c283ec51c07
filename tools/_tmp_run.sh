timeout 900 python3 -m pytest tests/test_mips_gpu.py tests/test_gradients_gpu.py -x -q -m gpu -k "overflowing or gradients or aux or golden or oracle or half or invalid" > gpurun_out/r2c_pytest.log 2>&1; echo rc=$?; tail -6 gpurun_out/r2c_pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c2trace -- python3 $GRAFT_REPO_ROOT/bench.py --rows 1000000 --nq 256 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/c2trace/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
idx=max(i for i,r in enumerate(rows) if "mips_prepare" in r["Kernel_Name"])
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx+8]:
    print("%8.1f %8.1f %8s %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r.get("Grid_Size_X",r.get("Grid_Size")),r["Kernel_Name"][:70]))
PY
