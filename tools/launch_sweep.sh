#!/bin/bash
# k_persist / k_rowstats / k_stats durations against the launch length (rocprofv3 kernel trace over tools/pmc_traffic.py: ten launches
# of S steps at the steady state): tools/launch_sweep.sh <tag> "<S list>" [workload] [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-sweep}
W=${3:-cfg2}; B=${4:-65536}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in $2; do
  rm -rf $O/s$S
  rocprofv3 --kernel-trace --output-format csv -d $O/s$S -- python3 $R/tools/pmc_traffic.py $W $B $S > $O/s$S.log 2>&1
  python3 - $O/s$S $S <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = {}
for name in ("k_persist", "k_rowstats", "k_stats"):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("void " + name)]
    d = d[-10:]
    out[name] = sorted(d)[len(d) // 2] / 1000.0 if d else 0.0
print("S=%s  k_persist %.1f us  k_rowstats %.1f us  k_stats %.1f us  sum %.1f" % (sys.argv[2], out["k_persist"], out["k_rowstats"], out["k_stats"], sum(out.values())))
PY
done
