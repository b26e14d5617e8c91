#!/bin/bash
# Dynamic instruction counts of k_persist for variants of the specialised kernel (diagnostic variants: ORL_SPEC_EXTRA="-DORL_DIAG -DORL_X_SKIP_..." , csrc/orl_diag.h) (deterministic, unlike timings: the pool's
# run-to-run spread is +-1.3 %).  usage (GPU box): tools/valu_ab.sh <tag> "<label>|<ENV=VAL;...>|<workload> <batch>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  IFS='|' read -r l envs wl <<< "$spec"
  set -- $wl
  rm -rf $O/$l
  (
    IFS=';' read -ra kv <<< "$envs"
    for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/$l -- python3 $R/tools/pmc_traffic.py $1 $2 128 > $O/$l.log 2>&1
  )
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/$l/**/*counter_collection.csv", recursive=True)
if not f:
    print("$l", "no counters:", open("$O/$l.log").read()[-300:])
else:
    rows = [r for r in csv.DictReader(open(f[0])) if "k_persist" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-10:]
    acc = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) in ids: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    nw = ($2 + 7) // 8
    per = lambda c: acc[c] / len(ids) / 128 / nw
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(glob.glob("$O/$l/**/*kernel_trace.csv", recursive=True)[0])) if "k_persist" in r["Kernel_Name"]][-10:]
    print("$l", "VALU %.1f SALU %.1f LDS %.1f VMEM_RD %.1f WR %.1f wait %.3f  us/step (under the profiler) %.2f" % (per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_INSTS_VMEM_RD"), per("SQ_INSTS_VMEM_WR"), acc["SQ_WAIT_ANY"] / max(acc["SQ_WAVE_CYCLES"], 1), sum(dur) / len(dur) / 128))
PY
done
