#!/usr/bin/env python3
"""Copy the judged summaries of tools/run_profiles.sh from gpurun_out/final into profiles/<tag>_* (newest files win)."""
import csv
import glob
import json
import os
import shutil
import sys

O = "gpurun_out/final"
tag = sys.argv[1]


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def mean_last(d, kern, n=20):
    f = newest("%s/%s/**/*counter_collection.csv" % (O, d))
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]][-n:]
    return sum(v) / len(v)


known = 65536 * 110 * 8
factor = known / (mean_last("tr_f", "k_calib", 4) * 1024)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_traffic.py), cfg2 B=65536, mean of the last 20 "
               "steady-state launches of each kernel; FETCH_SIZE x the factor measured on k_calib_read (known byte count; 8-B and 16-B per-lane "
               "loads both give 2.0, as MI355X_MICROARCH.md states); WRITE_SIZE as is",
       "workload": "cfg2", "batch": 65536, "fetch_calibration_factor": round(factor, 4), "kernels": {}}
# bench.py looks kernels up by these names
STEPS = {"k_persist": 20}  # tools/pmc_traffic.py measures launches of env.run(policy, 20)
for name, kern in (("k_persist", "k_persist<"), ("k_policy", "void k_policy<")):
    last = 5 if name in STEPS else 20  # (the run before the measured launches is one long launch of the same kernel)
    f = mean_last("tr_f", kern, last) * 1024 * factor
    w = mean_last("tr_w", kern, last) * 1024
    out["kernels"][name] = {"fetch_bytes_per_launch": int(f), "write_bytes_per_launch": int(w), "hbm_bytes_per_launch": int(f + w)}
    if name in STEPS:
        out["kernels"][name]["steps_per_launch"] = STEPS[name]
    try:  # L2 <-> fabric requests (TCC_EA0_RDREQ_sum + TCC_EA0_WRREQ_sum), the quantity the random-access roofline counts
        fe = newest("%s/ea/**/*counter_collection.csv" % O)
        rows = [r for r in csv.DictReader(open(fe)) if kern in r["Kernel_Name"]]
        rd = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "TCC_EA0_RDREQ_sum"][-last:]
        wr = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "TCC_EA0_WRREQ_sum"][-last:]
        out["kernels"][name]["dram_requests_per_launch"] = int(sum(rd) / len(rd) + sum(wr) / len(wr))
    except Exception as exc:  # noqa: BLE001
        print("no request counters for", name, exc)
json.dump(out, open("profiles/traffic_cfg2.json", "w"), indent=1)
shutil.copy(newest(O + "/stats/**/*kernel_stats.csv"), "profiles/%s_bench_cfg2_kernel_stats.csv" % tag)
shutil.copy(newest(O + "/stats1/**/*kernel_stats.csv"), "profiles/%s_bench_cfg2_single_stream_kernel_stats.csv" % tag)
for f in glob.glob(O + "/bench_*.json"):
    shutil.copy(f, "profiles/%s_%s" % (tag, os.path.basename(f)))
print(json.dumps(out["kernels"]))
# counter summaries (SQ issue/wait counters of the bench run, L2<->fabric request counts) as one text file
import subprocess
txt = subprocess.run([sys.executable, "tools/pmc_summary.py", O + "/sq", O + "/ea", O + "/tr_f", O + "/tr_w"], capture_output=True, text=True).stdout
open("profiles/%s_pmc_summary.txt" % tag, "w").write(txt.replace(os.getcwd() + "/", ""))
