#!/bin/bash
# quick A/B on the GPU box: usage tools/ab_quick.sh <tag> "<label> <workload> <batch> <steps> ENV=.. ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
for spec in "$@"; do
  set -- $spec
  l=$1; w=$2; b=$3; s=$4; shift 4
  env X=1 "$@" python3 $R/bench.py --gpus 1 --workload $w --batch $b --steps $s --no-cpu-baseline --min-timed-s 0.7 > $O/$l.json 2> $O/$l.err
  python3 -c "import json; d=json.load(open('$O/$l.json')); print('$l', d['value'], d['ms_per_step'], d['config']['step_kernels'])" || tail -3 $O/$l.err
done
