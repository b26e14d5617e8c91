import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["ORL_STEP_IMPL"] = "1"; os.environ["ORL_STREAMS"] = "1"
import optical_rl_gym_amd as orl
from bench import WORKLOADS
fam, topo, kw, policy = WORKLOADS["cfg2"]
B = 65536
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
for n in (200, 800, 500, 500):
    env.run(policy, n)
    print(n, "serial env-steps so far:", env.lib.orl_batch_debug_serial_count(env._h), "mean active", env.active().mean())
