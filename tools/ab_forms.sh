#!/bin/bash
# A/B on the GPU box: residency (rounds per launch) against per-wave speed.  usage: tools/ab_forms.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ab_forms}
mkdir -p $O
run() {  # label, workload, batch, steps, env...
  l=$1; w=$2; b=$3; s=$4; shift 4
  env "$@" python3 $R/bench.py --gpus 1 --workload $w --batch $b --steps $s --no-cpu-baseline --min-timed-s 0.7 > $O/$l.json 2> $O/$l.err
  python3 -c "import json; d=json.load(open('$O/$l.json')); print('$l', d['value'], d['ms_per_step'])" || tail -3 $O/$l.err
}
for w in cfg1 cfg3; do for s in 20 300; do for v in 4 5; do
  run ${w}_s${s}_v$v $w 65536 $s ORL_PERSIST_VARIANT=$v
done; done; done
for r in 12 11 10 9 8; do run cfg2_s20_r$r cfg2 65536 20 ORL_PERSIST_WGS_PER_CU=$r; done
for r in 12 11; do run cfg2_s300_r$r cfg2 65536 300 ORL_PERSIST_WGS_PER_CU=$r; done
run cfg2_s20_v5 cfg2 65536 20 ORL_PERSIST_VARIANT=5
run cfg2_s20_v0 cfg2 65536 20 ORL_PERSIST_VARIANT=0
run cfg2_b49152_s20 cfg2 49152 20 X=1
run cfg2_b24576_s20 cfg2 24576 20 X=1
run cfg2_b24576_s300 cfg2 24576 300 X=1
