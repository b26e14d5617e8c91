#!/bin/bash
# usage: tools/prof_cfg.sh <workload> <batch> <impl> : rocprofv3 per-kernel stats for one config (runs on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$1_$3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ORL_STEP_IMPL=$3 ORL_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload $1 --batch $2 --steps 200 --warmup 1500 --no-cpu-baseline > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-300
f=$(find $O -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-60s calls %7s avg %10.1f ns  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']), r['Percentage']))
PY
