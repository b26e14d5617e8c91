#!/bin/bash
# L2<->fabric traffic of k_persist for variants of the specialised kernel (FETCH_SIZE x 2, WRITE_SIZE: KiB per launch of 128 steps).
# usage (GPU box): tools/traffic_ab.sh <tag> "<label>|<ENV=VAL;...>|<workload> <batch>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  IFS='|' read -r l envs wl <<< "$spec"
  set -- $wl
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${l}_$c
    (
      IFS=';' read -ra kv <<< "$envs"
      for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
      timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${l}_$c -- python3 $R/tools/pmc_traffic.py $1 $2 128 > $O/${l}_$c.log 2>&1
    )
  done
  python3 - <<PY
import csv, glob
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$O/${l}_%s/**/*counter_collection.csv" % c, recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if "k_persist" in r["Kernel_Name"] and r["Counter_Name"] == c][-10:]
    out[c] = sum(float(r["Counter_Value"]) for r in rows) / len(rows) * 1024 / 128 / $2
print("$l", "fetch B/env-step %.0f (x2 calibration) write %.0f" % (2 * out["FETCH_SIZE"], out["WRITE_SIZE"]))
PY
done
