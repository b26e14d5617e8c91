"""Where OpticalVecEnv.step() spends its time at 65 536 envs (cProfile over 30 steps).  usage (GPU box): python tools/vec_env_prof.py [cfg2|cfg3] [f32]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import optical_rl_gym_amd as orl
from optical_rl_gym_amd.vec_env import OpticalVecEnv
from bench import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
fam, topo, kw, pol = WORKLOADS[name]
B = 65536
b = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
v = OpticalVecEnv(b, obs_dtype=np.float32 if "f32" in sys.argv else np.float64)
v.reset()
a = b.policy(pol)[:, :b.N_ACTION].copy() if fam != "DeepRMSA" else b.policy(pol)[:, 0].copy()
for _ in range(3):
    v.step(a)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    v.step(a)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
# the pieces, timed one by one
def t(f, n=30):
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3
print("step_async only (queue): %.3f ms" % t(lambda: (v.step_async(a), v.batch.step_wait())[0]))
print("batch.step(fetch=False) + sync: %.3f ms" % t(lambda: (b.step(a, auto_reset=True, fetch=False), b.sync())))
b.policy(pol, fetch=False)
print("batch.step(None, fetch=False) + sync: %.3f ms" % t(lambda: (b.step(None, auto_reset=True, fetch=False), b.sync())))
b.close()
