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
    if kern.startswith("k_persist"):
        # every launch of the persistent kernel is followed by k_stats, which replays the statistics log it wrote (deferred
        # bookkeeping, one lane per env): its traffic belongs to the launch
        try:
            fs = mean_last("tr_f" + suffix, "k_stats<", "FETCH_SIZE", last) * 1024 * factor
            ws = mean_last("tr_w" + suffix, "k_stats<", "WRITE_SIZE", last) * 1024
            rec["k_stats_bytes_per_launch"] = int(fs + ws)
            rec["hbm_bytes_per_launch"] = int(f + w + fs + ws)
            f, w = f + fs, w + ws
        except (FileNotFoundError, KeyError):
            pass
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


def launches_report(name, trace_csv, bench_log):
    """The timed blocks of a bench command inside its kernel trace.  The stats file averages over EVERY k_persist dispatch of the
    command — the 128-step launches of the state preparation, the timed launches, half-batch launches on two streams — so the
    dispatches are classified by what the command itself says it launched (its JSON line: blocks, launches per block): the timed
    blocks are the LAST blocks x launches-per-block x parts dispatches of k_persist, in start order (what follows them — the
    stand-alone scan and the host-driven loops — launches other kernels).  Per block: every dispatch's own duration, and the
    block's span on the device (first start to last end, k_stats — the deferred bookkeeping behind every launch — included:
    launches of two half batches overlap on two streams, so durations do not add up there).  The roofline fraction is recomputed
    from the spans and set beside the one the command measured with HIP events."""
    rows = [r for r in csv.DictReader(open(trace_csv)) if "k_persist" in r["Kernel_Name"] or "k_stats" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    pers = [r for r in rows if "k_persist" in r["Kernel_Name"]]
    if not pers:
        return None
    out = ["rocprofv3 --kernel-trace of the bench command for %s (tools/run_profiles.sh)" % name]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append("  all %d k_persist dispatches of the command (state preparation included): mean %.1f us, median %.1f us"
               % (len(pers), sum(map(dur, pers)) / len(pers), statistics.median(map(dur, pers))))
    bench = None
    try:
        bench = json.loads([ln for ln in open(bench_log) if ln.startswith("{")][-1])
    except Exception:
        pass
    if not bench:
        out.append("  (no bench line beside the trace: timed blocks not identified)")
        return "\n".join(out) + "\n"
    import re

    m = re.search(r"\((\d+) launches per (\d+)-step block\)", bench["config"]["step_kernels"][0])
    launches, steps = int(m.group(1)), int(m.group(2))
    blocks = bench["timing"]["blocks"]
    # a run of more than one launch on a batch of more than one generation is two half batches on two streams (orl_batch_run)
    # (Grid_Size_X counts work-items: a launch over the whole batch has ceil(B / 8) wavefronts of 64)
    envs = bench["config"].get("envs_per_gpu")
    envs = envs[0] if isinstance(envs, list) else envs
    tail = pers[-1]
    # (the two-wavefront form of small batches: workgroups of 128)
    full = ((int(envs) + 7) // 8) * int(tail.get("Workgroup_Size_X", 64) or 64)
    parts = 2 if int(tail.get("Grid_Size_X", full)) < full else 1
    per_block = launches * parts
    timed = pers[-blocks * per_block:] if blocks * per_block <= len(pers) else []
    if not timed:
        out.append("  (fewer k_persist dispatches than the bench line's %d blocks x %d: timed blocks not identified)" % (blocks, per_block))
        return "\n".join(out) + "\n"
    t_first = int(timed[0]["Start_Timestamp"])
    after = [r for r in rows if int(r["Start_Timestamp"]) >= t_first]
    spans, d_p, d_s = [], [], []
    for b in range(blocks):
        blk = timed[b * per_block:(b + 1) * per_block]
        lo = int(blk[0]["Start_Timestamp"])
        hi_lim = int(timed[(b + 1) * per_block]["Start_Timestamp"]) if (b + 1) * per_block < len(timed) else None
        mine = [r for r in after if int(r["Start_Timestamp"]) >= lo and (hi_lim is None or int(r["Start_Timestamp"]) < hi_lim)]
        if hi_lim is None:  # the last block: its k_persist dispatches and the k_stats right behind them
            last_end = max(int(r["End_Timestamp"]) for r in blk)
            mine = [r for r in mine if "k_persist" in r["Kernel_Name"] or int(r["Start_Timestamp"]) <= last_end + 200000][: 2 * per_block]
        spans.append((max(int(r["End_Timestamp"]) for r in mine) - lo) / 1e3)
        d_p += [dur(r) for r in mine if "k_persist" in r["Kernel_Name"]]
        d_s += [dur(r) for r in mine if "k_stats" in r["Kernel_Name"]]
    out.append("  timed blocks: the last %d blocks x %d k_persist dispatches (%d launch(es) of <= 128 steps per %d-step block%s)"
               % (blocks, per_block, launches, steps, ", two half batches on two streams" if parts == 2 else ""))
    out.append("    k_persist per dispatch: median %.1f us, mean %.1f, min %.1f, max %.1f" % (statistics.median(d_p), sum(d_p) / len(d_p), min(d_p), max(d_p)))
    if d_s:
        out.append("    k_stats   per dispatch: median %.1f us, mean %.1f (the bookkeeping of the launch in front of it, one lane per env)"
                   % (statistics.median(d_s), sum(d_s) / len(d_s)))
    sp = statistics.median(spans)
    out.append("    block span on the device (first start to last end): median %.1f us, mean %.1f, min %.1f, max %.1f"
               % (sp, sum(spans) / len(spans), min(spans), max(spans)))
    rf = bench["roofline"]
    bytes_block = rf["algorithmic_bytes_per_launch"] * launches
    frac_trace = bytes_block / (sp * 1e-6) / 1e9 / rf["peak"]
    out.append("    roofline from the trace: %d algorithmic bytes per block / %.1f us = %.1f GB/s = %.4f of %d GB/s"
               % (bytes_block, sp, bytes_block / (sp * 1e-6) / 1e9, frac_trace, rf["peak"]))
    out.append("    the same command's own line (HIP events; under the profiler): frac %.4f, %.1f us per launch x %d  ->  ratio trace / events %.3f"
               % (rf["frac"], rf["us_per_launch"], launches, frac_trace / rf["frac"] if rf["frac"] else 0.0))
    return "\n".join(out) + "\n"


src_hash = _build.source_hash(with_compiler=False)
# (workload, suffix of the pass directories, output file): the BASELINE sizes, and the literal 4 096-env batches of cfg1-3
for wl, sfx, outname in [(w, "", "traffic_%s.json" % w) for w in ("cfg2", "cfg1", "cfg3", "cfg4", "cfg5")] + \
                        [(w, "_b4096", "traffic_%s_4096.json" % w) for w in ("cfg2", "cfg1", "cfg3")]:
    log = os.path.join(O, "tr_f_%s%s.log" % (wl, sfx))
    if not os.path.exists(log):
        continue
    lines = [ln for ln in open(log) if ln.startswith("{")]
    if not lines:
        print(wl, "counter pass failed:", open(log).read()[-300:])
        continue
    run = json.loads(lines[-1])  # what tools/pmc_traffic.py printed
    factor = run["calibration_bytes"] / (mean_last("tr_f_" + wl + sfx, "k_calib", "FETCH_SIZE", 4) * 1024)
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum + TCC_EA0_WRREQ_sum / SQ_* (separate passes over "
                   "tools/pmc_traffic.py): mean of the measured launches; FETCH_SIZE (KiB) x the factor measured on k_calib_read's known "
                   "byte count (MI355X_MICROARCH.md: 2.0 for wide coalesced reads); WRITE_SIZE as is",
           "workload": run["workload"], "batch": run["batch"], "mean_active_services": run["mean_active_services"],
           "source_hash": src_hash, "tag": tag, "fetch_calibration_factor": round(factor, 4), "kernels": {}}
    out["kernels"]["k_persist"] = kernel_record("_" + wl + sfx, "k_persist<", run["measured_launches"], run["steps_per_launch"], factor)
    try:
        out["kernels"]["k_policy"] = kernel_record("_" + wl + sfx, "void k_policy<", 20, None, factor)
    except (FileNotFoundError, KeyError):
        pass
    log20 = os.path.join(O, "tr_f_%s_s20.log" % wl)
    if os.path.exists(log20) and not sfx:
        l20 = [ln for ln in open(log20) if ln.startswith("{")]
        if l20:
            r20 = json.loads(l20[-1])
            out["kernels"]["k_persist_steps%d" % r20["steps_per_launch"]] = kernel_record(
                "_%s_s20" % wl, "k_persist<", r20["measured_launches"], r20["steps_per_launch"], factor)
    json.dump(out, open(os.path.join(P, outname), "w"), indent=1)
    k = out["kernels"]["k_persist"]
    print(wl + sfx, "HBM bytes/step", k.get("hbm_bytes_per_step"), "VALU/wavefront-step",
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
        txt = launches_report(name, tr, os.path.join(O, "stats_%s.log" % name))
        if txt:
            open(os.path.join(P, "%s_bench_%s_launches.txt" % (tag, name)), "w").write(txt)
for f in glob.glob(O + "/bench_*.json") + glob.glob(O + "/phase_*.txt") + glob.glob(O + "/agent_loop_*.json") + glob.glob(O + "/pair_*.txt"):
    if os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, "%s_%s" % (tag, os.path.basename(f))))
passes = sorted(d for d in glob.glob(O + "/*") if os.path.isdir(d) and os.path.basename(d).split("_")[0] in ("sq1", "sq2", "sq3", "ea", "tr"))
txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + passes, capture_output=True, text=True).stdout
open(os.path.join(P, "%s_pmc_summary.txt" % tag), "w").write(txt.replace(ROOT + "/", ""))
print("profiles/%s_* written" % tag)
