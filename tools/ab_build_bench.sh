#!/bin/bash
# usage: tools/ab_build_bench.sh "<hipcc extra flags A>" "<flags B>" ...   (runs on the GPU box)
# Rebuilds liborlgpu.so with each flag set and prints value + per-kernel us for the cfg2 bench.
# ORL_HIPCC_EXTRA stays exported while the bench runs: the flags are part of the build stamp (_build.py).
for flags in "$@"; do
  export ORL_HIPCC_EXTRA="$flags"
  python -c "from optical_rl_gym_amd import _build; _build.build(force=True)" >/dev/null 2>&1
  echo "== flags: $flags"
  python bench.py --steps 300 --warmup 1500 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['us_per_launch'] for k,v in d['roofline_by_kernel'].items()})"
done
unset ORL_HIPCC_EXTRA
python -c "from optical_rl_gym_amd import _build; _build.build(force=True)" >/dev/null 2>&1
