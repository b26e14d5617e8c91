#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + SQ counters of the agent-in-the-loop step kernel (k_agent).
# usage: tools/agent_prof.sh <tag> [workload] [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-agent}; WL=${2:-cfg2}; B=${3:-65536}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats_agent $O/sq_agent $O/tr_agent
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_agent -- python3 $R/tools/agent_loop_rate.py $WL $B > $O/stats_agent.log 2>&1
f=$(find $O/stats_agent -name "*kernel_stats.csv" | head -1); head -6 $f
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/sq_agent -- python3 $R/tools/agent_loop_rate.py $WL $B > $O/sq_agent.log 2>&1
# (FETCH_SIZE and WRITE_SIZE in one pass abort rocprofv3 on this image: one pass each; every pass under its own timeout)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tr_agent -- python3 $R/tools/agent_loop_rate.py $WL $B > $O/tr_agent.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tw_agent -- python3 $R/tools/agent_loop_rate.py $WL $B > $O/tw_agent.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("sq_agent", "tr_agent", "tw_agent"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            if "k_agent" not in k and "k_policy" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add((k, r["Dispatch_Id"]))
        for k in acc:
            n = sum(1 for s in seen if s[0] == k)
            print(d, k, n, {c: "%.4g" % (v / n) for c, v in acc[k].items()})
PY
