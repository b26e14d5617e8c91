#!/bin/bash
# all BASELINE configs, wave64 monolith (64) vs split pipeline (1); runs on the GPU box
for cfg in "cfg2 65536" "cfg2 16384" "cfg2 8192" "cfg2 4096" "cfg1 4096" "cfg1 32768" "cfg3 4096" "cfg3 32768" "cfg4 16384" "cfg4n 16384" "cfg5 32768"; do
  set -- $cfg
  for impl in 64 1 2; do
    ORL_STEP_IMPL=$impl python bench.py --workload $1 --batch $2 --steps 200 --warmup 1500 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 B=$2 impl=$impl', '%.3e' % d['value'], d['ms_per_step'], d['state'])"
  done
done
