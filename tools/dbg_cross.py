"""Where do two step implementations differ on a batch?  usage: dbg_cross.py <implA> <implB> <workload> <batch> <steps>"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import optical_rl_gym_amd as orl
from bench import WORKLOADS
ia, ib, wl, B, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
fam, topo, kw, policy = WORKLOADS[wl]
kw = dict(kw, episode_length=90)
seeds = [77 + 3 * i for i in range(B)]
out = {}
for v in (ia, ib):
    os.environ["ORL_STEP_IMPL"] = v
    env = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, **kw)
    env.run(policy, steps)
    out[v] = (env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(),
              np.stack([env.net_stats(i) for i in range(min(B, 64))]), np.stack([env.link_stats(i) for i in range(min(B, 16))]),
              np.stack([env.slots(i) for i in range(min(B, 16))]), int(env.lib.orl_batch_debug_serial_count(env._h)))
    env.close()
a, b = out[ia], out[ib]
for k, name in enumerate(("counters", "services", "active", "flags", "net_stats", "link_stats", "slots")):
    x, y = np.asarray(a[k]), np.asarray(b[k])
    n = x.shape[0]
    d = (x != y).reshape(n, -1).any(axis=1)
    rows = np.where(d)[0]
    print(name, "differing envs:", len(rows), rows[:10])
    for r in rows[:2]:
        print("  env", r, ia, x[r].ravel()[:12], ib, y[r].ravel()[:12])
print("serial", a[7], b[7])
