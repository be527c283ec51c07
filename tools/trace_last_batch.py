"""Print start / duration of the kernels of the LAST search batch from a rocprofv3 --kernel-trace CSV directory."""
import csv
import glob
import sys

paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last_prepare = max(i for i, r in enumerate(rows) if "mips_prepare_kernel" in r[2])
t0 = rows[last_prepare][0]
for s, e, n in rows[last_prepare:]:
    if "mips_" not in n:
        continue
    print("%9.1f us  %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:70]))
