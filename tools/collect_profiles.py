#!/usr/bin/env python3
"""Copy the judged summaries of tools/run_profiles.sh from gpurun_out/<tag> into profiles/<tag>_* and derive
profiles/traffic_<workload>.json (HBM bytes and L2<->fabric requests per launch / per step of the persistent kernel and of
the stand-alone slot scan) from the counter passes.  The JSON carries the source hash of the build it was measured on:
bench.py reports `roofline.traffic` only when that matches the library it runs.

    python3 tools/collect_profiles.py <tag> [workload]
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optical_rl_gym_amd import _build  # noqa: E402

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
O = os.path.join(ROOT, "gpurun_out", tag)


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def mean_last(d, kern, counter, n):
    f = newest("%s/%s/**/*counter_collection.csv" % (O, d))
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == counter][-n:]
    return sum(v) / len(v)


run = json.loads([l for l in open(os.path.join(O, "tr_f.log")) if l.startswith("{")][-1])  # what tools/pmc_traffic.py printed
factor = run["calibration_bytes"] / (mean_last("tr_f", "k_calib", "FETCH_SIZE", 4) * 1024)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum + TCC_EA0_WRREQ_sum (separate passes over tools/pmc_traffic.py): "
               "mean of the measured launches; FETCH_SIZE (KiB) x the factor measured on k_calib_read's known byte count "
               "(MI355X_MICROARCH.md: 2.0 for wide coalesced reads); WRITE_SIZE as is",
       "workload": run["workload"], "batch": run["batch"], "mean_active_services": run["mean_active_services"],
       "source_hash": _build.source_hash(with_compiler=False), "tag": tag, "fetch_calibration_factor": round(factor, 4), "kernels": {}}
for name, kern, last, spl in (("k_persist", "k_persist<", run["measured_launches"], run["steps_per_launch"]), ("k_policy", "void k_policy<", 20, None)):
    f = mean_last("tr_f", kern, "FETCH_SIZE", last) * 1024 * factor
    w = mean_last("tr_w", kern, "WRITE_SIZE", last) * 1024
    rec = {"fetch_bytes_per_launch": int(f), "write_bytes_per_launch": int(w), "hbm_bytes_per_launch": int(f + w)}
    if spl:
        rec["steps_per_launch"] = spl
        rec["hbm_bytes_per_step"] = int((f + w) / spl)
    try:
        rq = mean_last("ea", kern, "TCC_EA0_RDREQ_sum", last) + mean_last("ea", kern, "TCC_EA0_WRREQ_sum", last)
        rec["dram_requests_per_launch"] = int(rq)
        if spl:
            rec["dram_requests_per_step"] = int(rq / spl)
    except Exception as exc:  # noqa: BLE001
        print("no request counters for", name, exc)
    # issue-side counters of the same launches (passes sq1..sq3 of tools/run_profiles.sh), per launch: what bounds a kernel
    # that moves few bytes (bench.py reports VALU-busy = SQ_ACTIVE_INST_VALU x 4 / (launch cycles x SIMDs) from these)
    sq = {}
    for d, names in (("sq1", ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                             "SQ_INSTS_VALU", "SQ_INSTS_SALU")),
                     ("sq2", ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")),
                     ("sq3", ("SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_BRANCH", "SQ_LDS_BANK_CONFLICT"))):
        for c in names:
            try:
                sq[c] = int(mean_last(d, kern, c, last))
            except Exception:  # noqa: BLE001  (pass not collected)
                pass
    if sq:
        rec["sq_per_launch"] = sq
    out["kernels"][name] = rec
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_%s.json" % workload), "w"), indent=1)
print(json.dumps(out["kernels"]))
shutil.copy(newest(O + "/stats/**/*kernel_stats.csv"), os.path.join(ROOT, "profiles", "%s_bench_cfg2_kernel_stats.csv" % tag))
if os.path.isdir(O + "/stats20"):
    shutil.copy(newest(O + "/stats20/**/*kernel_stats.csv"), os.path.join(ROOT, "profiles", "%s_bench_cfg2_steps20_kernel_stats.csv" % tag))
    # the stats file averages over EVERY k_persist dispatch of the command — the 128-step launches of the state preparation
    # (1 500 steps) and the warm-up included; the timed blocks are the 20-step launches: their durations from the kernel trace
    import csv
    import statistics

    dur = []
    for r in csv.DictReader(open(newest(O + "/stats20/**/*kernel_trace.csv"))):
        if "k_persist" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if dur:
        cut = 2.5 * statistics.median(dur)
        short, long_ = [d for d in dur if d <= cut], [d for d in dur if d > cut]
        with open(os.path.join(ROOT, "profiles", "%s_bench_cfg2_steps20_launches.txt" % tag), "w") as f:
            f.write("rocprofv3 --kernel-trace of `bench.py --gpus 1 --steps 20 --warmup 5` (tools/run_profiles.sh), k_persist dispatches:\n")
            f.write("  %d launches of 20 steps (warm-up + timed blocks): median %.1f us, mean %.1f us, min %.1f, max %.1f\n"
                    % (len(short), statistics.median(short), sum(short) / len(short), min(short), max(short)))
            if long_:
                f.write("  %d launches of up to 128 steps (state preparation: 1 500 steps before the timed region): mean %.1f us\n"
                        % (len(long_), sum(long_) / len(long_)))
            f.write("  (the *_kernel_stats.csv average of %.1f us is over all %d dispatches; bench.py's roofline.us_per_launch is the HIP-event\n"
                    "   time of the timed 20-step launches)\n" % (sum(dur) / len(dur), len(dur)))
for f in glob.glob(O + "/bench_*.json") + glob.glob(O + "/phase_*.txt") + glob.glob(O + "/agent_loop_*.json"):
    shutil.copy(f, os.path.join(ROOT, "profiles", "%s_%s" % (tag, os.path.basename(f))))
passes = [os.path.join(O, d) for d in ("sq1", "sq2", "sq3", "hit", "ea", "tr_f", "tr_w") if os.path.isdir(os.path.join(O, d))]
txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + passes, capture_output=True, text=True).stdout
open(os.path.join(ROOT, "profiles", "%s_pmc_summary.txt" % tag), "w").write(txt.replace(ROOT + "/", ""))
