"""Event counts of the pending-release push (needs a -DORL_TIMING=3 build)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["ORL_STEP_IMPL"] = "1"; os.environ["ORL_STREAMS"] = "1"
import numpy as np
import optical_rl_gym_amd as orl
from bench import WORKLOADS
fam, topo, kw, policy = WORKLOADS["cfg2"]
B = 65536
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
env.run(policy, 1500)
out = np.zeros(64, np.uint64)
env.lib.orl_batch_debug_prof(env._h, out.ctypes.data, 1)
env.run(policy, 100)
env.lib.orl_batch_debug_prof(env._h, out.ctypes.data, 1)
o = out.astype(float)
print("pushes %d; without hint %.3f; mean hwm %.1f; mean pending %.1f; scan windows per hint-less push %.2f; soon-list inserts %.3f"
      % (o[0], o[1] / o[0], o[2] / o[0], o[3] / o[0], o[4] / max(o[1], 1), o[5] / o[0]))
print("deferrals: empty-list %d (clock past horizon %d, something due %d), capacity %d" % (o[8], o[9], o[10], o[11]))
print("rebuilds with T <= now and something due:", int(out[12]))
f = out[16:].view(np.float64)
print("now %.6f T %.6f hwm %d nd %d" % (f[0], f[1], out[18], out[19]))
for l in range(8):
    print("lane", l, "kb[NS] q", int(out[20 + l] >> 40), "ord", int((out[20 + l] >> 32) & 255), "kb0 q", int((out[20 + l] & 0xffffffff) >> 8),
          "kb1 q", int(out[36 + l] >> 40), "kb2 q", int((out[36 + l] & 0xffffffff) >> 8), "T_lane", f[12 + l], "bt0", f[28 + l])
