#!/bin/bash
# A/B on the GPU box: how a short device-resident run (the driver's --steps 20 block) should be cut into launches.
# usage: tools/ab_short_runs.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ab_short}
mkdir -p $O
run() {  # label, env...
  l=$1; shift
  env "$@" python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --min-timed-s 1 > $O/$l.json 2> $O/$l.err
  python3 -c "import json; d=json.load(open('$O/$l.json')); print('$l', d['value'], d['ms_per_step'], d['timing']['block_s_min'], d['timing']['block_s_max'])"
}
run base X=1
run c10p2 ORL_PERSIST_CHUNK=10 ORL_PERSIST_PARTS=2
run c5p2 ORL_PERSIST_CHUNK=5 ORL_PERSIST_PARTS=2
run c7p2 ORL_PERSIST_CHUNK=7 ORL_PERSIST_PARTS=2
run c4p2 ORL_PERSIST_CHUNK=4 ORL_PERSIST_PARTS=2
run c10p1 ORL_PERSIST_CHUNK=10 ORL_PERSIST_PARTS=1
run c20p2 ORL_PERSIST_CHUNK=20 ORL_PERSIST_PARTS=2
for s in 64 100 300; do
  python3 $R/bench.py --gpus 1 --steps $s --no-cpu-baseline --min-timed-s 1 > $O/s$s.json 2> $O/s$s.err
  python3 -c "import json; d=json.load(open('$O/s$s.json')); print('steps $s', d['value'], d['ms_per_step'])"
done
