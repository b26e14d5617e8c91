import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_parity as T
import optical_rl_gym_amd as orl
from bench import WORKLOADS
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
chunks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "130,97,173").split(",")]
fam, topo, kw, policy = WORKLOADS[workload]
kw = dict(kw, episode_length=70)
seeds = [5 + 11 * i for i in range(batch)]
out = {}
for name, masks in (("wave64", None), ("persist", os.environ.get("MASKS", "1"))):
    for k, val in T.IMPL_ENV[name].items():
        if val is None: os.environ.pop(k, None)
        else: os.environ[k] = val
    if masks:
        os.environ["ORL_ITEM_MASKS"] = masks; os.environ["ORL_JIT_SPEC"] = "1"
    else:
        os.environ.pop("ORL_ITEM_MASKS", None)
    env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
    for c in chunks:
        env.run(policy, c)
    print(name, "form", int(env.lib.orl_batch_debug_persist_form(env._h)), "spec", int(env.lib.orl_batch_debug_persist_spec(env._h)), "serial", int(env.lib.orl_batch_debug_serial_count(env._h)))
    out[name] = dict(counters=env.counters().copy(), services=env.services().copy(), net=env.net_stats_all().copy(), link=env.link_stats_all().copy(), slots=env.slots_packed().copy())
    env.close()
for key in out["wave64"]:
    a, b = out["persist"][key], out["wave64"][key]
    d = np.argwhere(np.asarray(a) != np.asarray(b))
    print(key, "differs at", len(d), "entries; first", d[:5].tolist())
    if key == "net" and len(d):
        e = d[0][0]
        print(" env", e, "got", a[e], "exp", b[e])
        print(" columns differing:", sorted(set(int(x[1]) for x in d)), "envs", len(set(int(x[0]) for x in d)))
