import os, sys
root = sys.argv[5] if len(sys.argv) > 5 else os.getcwd()
os.chdir(root)
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_parity as T
import optical_rl_gym_amd as orl
from bench import WORKLOADS
v, masks, workload, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
for k, val in T.IMPL_ENV[v].items():
    if val is None: os.environ.pop(k, None)
    else: os.environ[k] = val
if masks != "-": os.environ["ORL_ITEM_MASKS"] = masks
fam, topo, kw, policy = WORKLOADS[workload]
kw = dict(kw, episode_length=70)
env = orl.make(fam, topology=topo, num_envs=batch, seeds=[5 + 11 * i for i in range(batch)], **kw)
n = int(os.environ.get("RUN_STEPS", "500"))
env.run(policy, n)
print("ok", orl.__file__, v, masks, workload, flush=True)
