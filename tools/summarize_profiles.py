#!/usr/bin/env python3
"""Summarise gpurun_out/<round>/<workload>_* (tools/profile_r3.sh) into profiles/<round>_<workload>_*.

For every workload: the bench line, the rocprofv3 kernel stats CSV, a per-step kernel timeline (from the kernel trace)
and one JSON with the PMC sums over the filter launches of ONE batch: HBM traffic = FETCH_SIZE * 1024 * 2 (gfx950 reports
half of wide streaming reads, MI355X_MICROARCH.md "HBM") + WRITE_SIZE * 1024; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / CUs*4
over GRBM_GUI_ACTIVE / 8 XCDs.  Also refreshes profiles/hbm_traffic.json, which bench.py quotes as `roofline.traffic`."""
import collections
import csv
import glob
import json
import pathlib
import shutil
import sys

rnd = sys.argv[1]  # e.g. r02
src = pathlib.Path("gpurun_out") / rnd
prof = pathlib.Path("profiles")
traffic_file = prof / "hbm_traffic.json"
traffic = json.loads(traffic_file.read_text()) if traffic_file.exists() else {}
traffic = {k: v for k, v in traffic.items() if isinstance(v, dict)}  # drop pre-round-2 bare numbers
for bench in sorted(src.glob("*_bench.json")):
    w = bench.name[: -len("_bench.json")]
    txt = [ln for ln in bench.read_text().splitlines() if ln.startswith("{")]
    if not txt:
        print(w, "no bench line")
        continue
    line = json.loads(txt[-1])
    (prof / f"{rnd}_{w}_bench.json").write_text(json.dumps(line) + "\n")
    r = line["roofline"]
    print(f"{w}: {line['value']:.0f} q/s, {line['ms_per_step']:.4f} ms/step, filter {r['kernel_ms_per_step']:.4f} ms in {r['launches_per_step']:.0f} launches, "
          f"{r['bound']} {r['achieved']:.1f} {r['unit']} = {r['frac']:.3f}, verify {line.get('verify')}")
    stats = glob.glob(str(src / f"{w}_prof" / "*" / "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], prof / f"{rnd}_{w}_kernel_stats.csv")
        for row in list(csv.DictReader(open(stats[0])))[:4]:
            print("   ", row["Name"][:70], row["Calls"], row["AverageNs"], row["Percentage"])
    trace = glob.glob(str(src / f"{w}_prof" / "*" / "*_kernel_trace.csv"))
    if trace:  # kernels of the last batch, in time order
        rows = sorted(csv.DictReader(open(trace[0])), key=lambda x: int(x["Start_Timestamp"]))
        names = [x["Kernel_Name"] for x in rows]
        last_prepare = max(i for i, nm in enumerate(names) if "mips_prepare" in nm)
        sel = rows[last_prepare:]
        t0 = int(sel[0]["Start_Timestamp"])
        with open(prof / f"{rnd}_{w}_last_batch_timeline.csv", "w") as f:
            f.write("start_us,duration_us,grid,kernel\n")
            for x in sel:
                if "mips_" not in x["Kernel_Name"] and "merge_topk" not in x["Kernel_Name"] and "exact_rescore" not in x["Kernel_Name"]:
                    continue
                f.write(f"{(int(x['Start_Timestamp']) - t0) / 1e3:.1f},{(int(x['End_Timestamp']) - int(x['Start_Timestamp'])) / 1e3:.1f},"
                        f"{x.get('Grid_Size_X', x.get('Grid_Size', ''))},\"{x['Kernel_Name'][:90]}\"\n")
    tot = collections.defaultdict(float)
    n_launch = int(round(r["launches_per_step"]))
    ok = True
    for d in ["sq1", "tcc1", "tcc2"]:
        files = glob.glob(str(src / f"{w}_pmc_{d}" / "*" / "*_counter_collection.csv"))
        if not files:
            ok = False
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(files[0])):
            if "mips_filter" in row["Kernel_Name"]:
                per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        for i in sorted(per)[-n_launch:]:  # the filter launches of the last batch
            for k, v in per[i].items():
                tot[k] += v
    if not ok or not tot:
        continue
    s = dict(tot)
    fetch, wr = tot["FETCH_SIZE"] * 1024 * 2, tot["WRITE_SIZE"] * 1024
    s.update(filter_launches_per_step=n_launch, hbm_read_bytes_corrected=fetch, hbm_write_bytes=wr,
             hbm_traffic_bytes_per_step=fetch + wr, algorithmic_bytes_per_step=r["algorithmic_bytes_per_step"],
             traffic_over_algorithmic=(fetch + wr) / r["algorithmic_bytes_per_step"],
             mfma_busy_frac=tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (tot["GRBM_GUI_ACTIVE"] / 8) if tot.get("GRBM_GUI_ACTIVE") else None,
             l2_hit_rate=tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]) if tot.get("TCC_HIT_sum") else None,
             workload=line["config"]["workload"],
             note="FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide streaming reads); separate --pmc passes "
                  "sq1/tcc1/tcc2; sums over the filter launches (bootstrap + stages) of one batch")
    (prof / f"{rnd}_{w}_pmc_filter_per_step.json").write_text(json.dumps(s, indent=1))
    cfg = line["config"]["workload"].split()
    key = (f"{cfg[0]}x{cfg[3]}x{cfg[6]}@{line['n_gpus']}" + ("" if line["data"] == "synthetic" else "/clustered")
           + ("/exact" if "exact-f32" in line["config"]["workload"] else ""))  # bench.py's traffic_key(): the store mode is part of the key
    traffic[key] = {"bytes": fetch + wr, "source": f"profiles/{rnd}_{w}_pmc_filter_per_step.json"}
    print(f"    PMC: HBM {fetch + wr:.4g} B = {s['traffic_over_algorithmic']:.3f}x algorithmic, MFMA busy {s['mfma_busy_frac']}, L2 hit {s['l2_hit_rate']}, "
          f"LDS conflicts {tot.get('SQ_LDS_BANK_CONFLICT')}")
traffic_file.write_text(json.dumps(traffic, indent=1))
