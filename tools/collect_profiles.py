#!/usr/bin/env python3
"""Copy the judged summaries of tools/run_profiles.sh from gpurun_out/<tag> into profiles/<tag>_* and derive
profiles/traffic_<workload>.json for every workload whose counter passes are there: HBM bytes, L2<->fabric requests and the SQ
issue counters per launch / per step of the persistent kernel (ten 128-step launches; for cfg2 also ten 20-step launches, the
shape of the driver's `bench.py --steps 20` blocks: `k_persist_steps20`) and of the stand-alone slot scan.  Each JSON carries
the source hash of the build it was measured on: bench.py reports `roofline.traffic` / `roofline.valu` only when that matches
the library it runs.

    python3 tools/collect_profiles.py <tag>
"""
import csv
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optical_rl_gym_amd import _build  # noqa: E402

tag = sys.argv[1]
O = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def newest(pattern):
    found = glob.glob(pattern, recursive=True)
    return max(found, key=os.path.getmtime) if found else None


def mean_last(d, kern, counter, n):
    f = newest("%s/%s/**/*counter_collection.csv" % (O, d))
    if f is None:
        raise FileNotFoundError(d)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == counter][-n:]
    if not v:
        raise KeyError(counter)
    return sum(v) / len(v)


SQ = (("sq1", ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
               "SQ_INSTS_VALU", "SQ_INSTS_SALU")),
      ("sq2", ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")),
      ("sq3", ("SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64",
               "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_BRANCH", "SQ_LDS_BANK_CONFLICT")))


def kernel_record(suffix, kern, last, spl, factor):
    f = mean_last("tr_f" + suffix, kern, "FETCH_SIZE", last) * 1024 * factor
    w = mean_last("tr_w" + suffix, kern, "WRITE_SIZE", last) * 1024
    rec = {"fetch_bytes_per_launch": int(f), "write_bytes_per_launch": int(w), "hbm_bytes_per_launch": int(f + w)}
    if spl:
        rec["steps_per_launch"] = spl
        rec["hbm_bytes_per_step"] = int((f + w) / spl)
    try:
        rq = mean_last("ea" + suffix, kern, "TCC_EA0_RDREQ_sum", last) + mean_last("ea" + suffix, kern, "TCC_EA0_WRREQ_sum", last)
        rec["dram_requests_per_launch"] = int(rq)
        if spl:
            rec["dram_requests_per_step"] = int(rq / spl)
    except (FileNotFoundError, KeyError):
        pass
    sq = {}
    for d, names in SQ:
        for c in names:
            try:
                sq[c] = int(mean_last(d + suffix, kern, c, last))
            except (FileNotFoundError, KeyError):
                pass
    if sq:
        rec["sq_per_launch"] = sq
    return rec


src_hash = _build.source_hash(with_compiler=False)
for wl in ("cfg2", "cfg1", "cfg3", "cfg4", "cfg5"):
    log = os.path.join(O, "tr_f_%s.log" % wl)
    if not os.path.exists(log):
        continue
    lines = [ln for ln in open(log) if ln.startswith("{")]
    if not lines:
        print(wl, "counter pass failed:", open(log).read()[-300:])
        continue
    run = json.loads(lines[-1])  # what tools/pmc_traffic.py printed
    factor = run["calibration_bytes"] / (mean_last("tr_f_" + wl, "k_calib", "FETCH_SIZE", 4) * 1024)
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum + TCC_EA0_WRREQ_sum / SQ_* (separate passes over "
                   "tools/pmc_traffic.py): mean of the measured launches; FETCH_SIZE (KiB) x the factor measured on k_calib_read's known "
                   "byte count (MI355X_MICROARCH.md: 2.0 for wide coalesced reads); WRITE_SIZE as is",
           "workload": run["workload"], "batch": run["batch"], "mean_active_services": run["mean_active_services"],
           "source_hash": src_hash, "tag": tag, "fetch_calibration_factor": round(factor, 4), "kernels": {}}
    out["kernels"]["k_persist"] = kernel_record("_" + wl, "k_persist<", run["measured_launches"], run["steps_per_launch"], factor)
    try:
        out["kernels"]["k_policy"] = kernel_record("_" + wl, "void k_policy<", 20, None, factor)
    except (FileNotFoundError, KeyError):
        pass
    log20 = os.path.join(O, "tr_f_%s_s20.log" % wl)
    if os.path.exists(log20):
        l20 = [ln for ln in open(log20) if ln.startswith("{")]
        if l20:
            r20 = json.loads(l20[-1])
            out["kernels"]["k_persist_steps%d" % r20["steps_per_launch"]] = kernel_record(
                "_%s_s20" % wl, "k_persist<", r20["measured_launches"], r20["steps_per_launch"], factor)
    json.dump(out, open(os.path.join(P, "traffic_%s.json" % wl), "w"), indent=1)
    k = out["kernels"]["k_persist"]
    print(wl, "HBM bytes/step", k.get("hbm_bytes_per_step"), "VALU/wavefront-step",
          round(k.get("sq_per_launch", {}).get("SQ_INSTS_VALU", 0) / k["steps_per_launch"] / ((run["batch"] + 7) // 8), 1))

# kernel traces of the bench commands
for d in sorted(glob.glob(O + "/stats_*")):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len("stats_"):]
    st = newest(d + "/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(P, "%s_bench_%s_kernel_stats.csv" % (tag, name)))
    tr = newest(d + "/**/*kernel_trace.csv")
    if tr:
        # the stats file averages over EVERY k_persist dispatch of the command — the 128-step launches of the state preparation
        # included; the timed blocks are the launches of --steps steps: their durations from the kernel trace
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr)) if "k_persist" in r["Kernel_Name"]]
        if dur:
            with open(os.path.join(P, "%s_bench_%s_launches.txt" % (tag, name)), "w") as f:
                f.write("rocprofv3 --kernel-trace of the bench command for %s (tools/run_profiles.sh), k_persist dispatches by duration:\n" % name)
                groups = {}
                for x in dur:
                    groups.setdefault(round(x / (0.15 * statistics.median(dur) + 1e-9)), []).append(x)
                med = statistics.median(dur)
                short = [x for x in dur if x <= 1.6 * min(dur)]
                f.write("  all %d dispatches: mean %.1f us, median %.1f us\n" % (len(dur), sum(dur) / len(dur), med))
                f.write("  the %d shortest-class dispatches (within 1.6x of the minimum: the timed blocks of a --steps 20 command, or the\n"
                        "  half-batch launches of a longer one): median %.1f us, mean %.1f us, min %.1f, max %.1f\n"
                        % (len(short), statistics.median(short), sum(short) / len(short), min(short), max(short)))
for f in glob.glob(O + "/bench_*.json") + glob.glob(O + "/phase_*.txt") + glob.glob(O + "/agent_loop_*.json"):
    if os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, "%s_%s" % (tag, os.path.basename(f))))
passes = sorted(d for d in glob.glob(O + "/*") if os.path.isdir(d) and os.path.basename(d).split("_")[0] in ("sq1", "sq2", "sq3", "ea", "tr"))
txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + passes, capture_output=True, text=True).stdout
open(os.path.join(P, "%s_pmc_summary.txt" % tag), "w").write(txt.replace(ROOT + "/", ""))
print("profiles/%s_* written" % tag)
