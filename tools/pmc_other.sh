#!/bin/bash
# Runs on the GPU box: the SQ issue counters and the HBM traffic of k_persist for one more workload (tools/run_profiles.sh
# does cfg2).  usage: tools/pmc_other.sh <tag> <workload> <batch>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; WL=$2; B=$3
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  n=$1; shift
  rm -rf $O/${n}_$WL
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/${n}_$WL -- python3 $R/tools/pmc_traffic.py $WL $B > $O/${n}_$WL.log 2>&1
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT
pass tr_f FETCH_SIZE
pass tr_w WRITE_SIZE
python3 - <<PY
import csv, glob, collections
for d in ("sq1", "sq2", "tr_f", "tr_w"):
    for f in glob.glob("$O/%s_$WL/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float); ids = []
        for r in csv.DictReader(open(f)):
            if "k_persist" not in r["Kernel_Name"]: continue
            ids.append(int(r["Dispatch_Id"]))
        last = sorted(set(ids))[-10:]
        for r in csv.DictReader(open(f)):
            if "k_persist" in r["Kernel_Name"] and int(r["Dispatch_Id"]) in last: acc[r["Counter_Name"]] += float(r["Counter_Value"])
        print("$WL", d, "per launch (mean of last %d):" % len(last), {c: "%.4g" % (v / max(1, len(last))) for c, v in acc.items()})
PY
