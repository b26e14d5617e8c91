import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_parity as T
import optical_rl_gym_amd as orl
from bench import WORKLOADS
workload, batch = "cfg5", 4096
fam, topo, kw, policy = WORKLOADS[workload]
kw = dict(kw, episode_length=70)
seeds = [5 + 11 * i for i in range(batch)]
out = {}
for label, name, extra in (("wave64", "wave64", {}), ("pair_m1", "persist", dict(ORL_ITEM_MASKS="1", ORL_JIT_SPEC="1")), ("pair", "persist", dict(ORL_JIT_SPEC="1")),
                           ("single_m1", "persist", dict(ORL_ITEM_MASKS="1", ORL_JIT_SPEC="1", ORL_PERSIST_RW="0")), ("global_m1", "persist_global", dict(ORL_ITEM_MASKS="1"))):
    for k, val in T.IMPL_ENV[name].items():
        if val is None: os.environ.pop(k, None)
        else: os.environ[k] = val
    os.environ.pop("ORL_ITEM_MASKS", None)
    for k, v in extra.items(): os.environ[k] = v
    env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
    env.run(policy, 130)
    print(label, "form", int(env.lib.orl_batch_debug_persist_form(env._h)), "spec", int(env.lib.orl_batch_debug_persist_spec(env._h)), "serial", int(env.lib.orl_batch_debug_serial_count(env._h)), flush=True)
    out[label] = env.net_stats_all().copy()
    env.close()
ref = out["wave64"]
for label, a in out.items():
    d = np.argwhere(a != ref)
    print(label, "vs wave64: differs at", len(d), "envs", sorted(set(int(x[0]) for x in d))[:30])
