#!/bin/bash
# usage: tools/ab_prof.sh "<hipcc extra flags A>" "<flags B>" ...   (runs on the GPU box)
# per-kernel rocprofv3 averages of the split pipeline (cfg2, one stream) for each flag set
R=${GRAFT_REPO_ROOT:-$(pwd)}
for flags in "$@"; do
  export ORL_HIPCC_EXTRA="$flags"
  python -c "from optical_rl_gym_amd import _build; _build.build(force=True)" >/dev/null 2>&1
  echo "== flags: $flags"
  $R/tools/prof_cfg.sh cfg2 65536 1 | grep -v "k_ctrl_a<\|k_ctrl_b1\|k_policy<\|k_seed\|copyBuffer\|k_reset\|k_init\|k_totals\|fillBuffer"
  cd $R
done
unset ORL_HIPCC_EXTRA
python -c "from optical_rl_gym_amd import _build; _build.build(force=True)" >/dev/null 2>&1
