#!/usr/bin/env python3
"""Per-wavefront clocks of k_agent (diagnostic build liborlgpu_timing.so): how long a wavefront of the agent's step kernel lives, and what
the ones that rebuild their envs' soon lists in that step add to the launch.   python3 tools/agent_timeline.py [workload] [batch]"""
import math
import os
import sys

os.environ["ORL_LIB_VARIANT"] = "timing"
os.environ["ORL_JIT_SPEC"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS, workload_load  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
fam, topo, kw, policy = WORKLOADS[name]
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
env.run(policy, max(1500, int(math.ceil(5 * workload_load(kw)))))
for _ in range(20):
    env.policy_step(policy, auto_reset=True, fetch=False)
env.sync()
ts = np.zeros(16384 * 8, np.uint64)
env._ck(env.lib.orl_batch_debug_prof(env._h, ts.ctypes.data, 2))
waves = (B + 7) // 8
t = ts.reshape(-1, 8)[:waves].astype(np.int64)
t0 = t[:, 0].min()
ent, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
dur = end - ent
reb = t[:, 2] > 0
print("%s B=%d: k_agent (scan fused), %d wavefronts; last one ends %.1f us after the first enters" % (name, B, waves, end.max()))
print("lifetime of a wavefront: mean %.1f us (sd %.1f); %d wavefronts (%.1f %%) rebuilt their soon lists in this step: %.1f us (sd %.1f), the others %.1f us (sd %.1f)"
      % (dur.mean(), dur.std(), reb.sum(), 100.0 * reb.mean(), dur[reb].mean() if reb.any() else 0, dur[reb].std() if reb.any() else 0,
         dur[~reb].mean(), dur[~reb].std()))
g1 = ent < 5.0
print("first generation: %d wavefronts, end at %s" % (g1.sum(), " ".join("%.1f" % v for v in np.percentile(end[g1], [1, 50, 90, 99, 100]))))
print("second generation: entry %s; end %s" % (" ".join("%.1f" % v for v in np.percentile(ent[~g1], [0, 50, 90, 99, 100])) if (~g1).any() else "-",
                                               " ".join("%.1f" % v for v in np.percentile(end[~g1], [1, 50, 90, 99, 100])) if (~g1).any() else "-"))
print("rebuilding wavefronts: %.1f us in the rebuild scan, %.1f us in the release loop (others: %.1f)" %
      (t[reb, 2].mean() / 2400.0, t[reb, 3].mean() / 2400.0, t[~reb, 3].mean() / 2400.0))
late = end > np.percentile(end, 99)
print("the last 1 %% of the wavefronts to end: %.0f %% of them rebuilt; lifetime %.1f us" % (100.0 * reb[late].mean(), dur[late].mean()))
env.close()
