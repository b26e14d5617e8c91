#!/bin/bash
# agent-loop A/B on the GPU box: tools/ab_agent.sh <tag> "<label>|<ORL_SPEC_EXTRA>|<workload>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
for spec in "$@"; do
  IFS='|' read -r l x w <<< "$spec"
  ORL_SPEC_EXTRA="$x" python3 $R/tools/agent_loop_rate.py $w 65536 2>/dev/null | grep "^{" > $O/$l.json
  python3 -c "import json; d=json.load(open('$O/$l.json')); print('$l', d['step_only'], d['policy_and_step'], d['host_driven_pcie'])"
done
