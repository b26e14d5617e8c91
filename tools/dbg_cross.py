"""Where do the two step implementations differ on a full-size batch?  (debug aid)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import optical_rl_gym_amd as orl
from bench import WORKLOADS
fam, topo, kw, policy = WORKLOADS["cfg2"]
kw = dict(kw, episode_length=90)
B = 65536
seeds = [77 + 3 * i for i in range(B)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 700
out = {}
for v in ("64", "1"):
    os.environ["ORL_STEP_IMPL"] = v
    env = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, **kw)
    env.run(policy, steps)
    out[v] = (env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(),
              int(env.lib.orl_batch_debug_serial_count(env._h)))
    env.close()
a, b = out["64"], out["1"]
for k, name in enumerate(("counters", "services", "active", "flags")):
    x, y = np.asarray(a[k]), np.asarray(b[k])
    d = (x != y)
    rows = np.where(d.reshape(B, -1).any(axis=1))[0]
    print(name, "differing envs:", len(rows), rows[:10])
    for r in rows[:3]:
        print("  env", r, "wave64", x[r], "split", y[r])
print("serial", a[4], b[4])
