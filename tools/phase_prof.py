"""Per-phase shader-clock breakdown of the pipeline kernels.  Build with ORL_HIPCC_EXTRA=-DORL_TIMING=<n> and run with
the same variable exported: n = 2 -> k_step_a2 / k_policy_ctrl_a, 4 -> the row kernels (default ORL_STEP_IMPL=2);
n = 1 -> k_ctrl_b2 (run with ORL_STEP_IMPL=1); n = 3 -> event counts, see tools/push_dbg.py."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("ORL_STEP_IMPL", "2"); os.environ["ORL_STREAMS"] = "1"
import numpy as np
import optical_rl_gym_amd as orl
from bench import WORKLOADS
fam, topo, kw, policy = WORKLOADS["cfg2"]
B = 65536
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
env.run(policy, 1500)
out = np.zeros(32, np.uint64)
env.lib.orl_batch_debug_prof(env._h, out.ctypes.data, 1)
n = 200
env.run(policy, n)
env.lib.orl_batch_debug_prof(env._h, out.ctypes.data, 1)
waves = B / 8 * n
for k in range(14):
    if out[k]:
        print("phase %2d: %9.0f cycles per wavefront" % (k, out[k] / waves))
for k in range(14):
    if out[16 + k]:
        print("slow wavefronts, phase %2d: %9.0f cycles" % (k, out[16 + k] / max(int(out[14]), 1)))
print("wavefront launches over 60k cycles: %d of %d; slowest: %d cycles" % (out[14], waves, out[15]))
