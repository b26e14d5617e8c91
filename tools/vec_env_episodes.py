"""OpticalVecEnv.step() at the steady state of cfg3 (DeepRMSA, 50-step episodes, 65 536 envs): ~1 300 envs finish an episode per step,
each of which SB3's VecEnv contract gives an info dict of its own.  usage (GPU box): python tools/vec_env_episodes.py [profile]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS  # noqa: E402
from optical_rl_gym_amd.vec_env import OpticalVecEnv  # noqa: E402

fam, topo, kw, pol = WORKLOADS["cfg3"]
B = 65536
b = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
v = OpticalVecEnv(b, obs_dtype=np.float32)
v.reset()
a = b.policy(pol)[:, 0].copy()
for _ in range(60):
    v.step(a)
t0 = time.perf_counter()
n, fin, looked = 100, 0, 0
for _ in range(n):
    o, r, d, i = v.step(a)
    fin += int(d.sum())
dt = time.perf_counter() - t0
print("cfg3 (episode_length 50) VecEnv.step f32: %.2f ms per step, %d envs finish an episode per step" % (dt / n * 1e3, fin / n))
# ... and with a consumer that looks at every finished env's episode row, as SB3's Monitor statistics do
t0 = time.perf_counter()
for _ in range(n):
    o, r, d, i = v.step(a)
    for e in np.flatnonzero(d).tolist():
        looked += 1 if i[e].get("episode") is not None else 0
dt2 = time.perf_counter() - t0
print("  with every finished env's info read: %.2f ms per step (%d rows read); episodes logged %d, kept %d" % (dt2 / n * 1e3, looked, v.episode_log.total, len(v.episode_log)))
if v._timing:
    n_s = v._timing.pop("steps")
    print("  sections, ms per step:", {k: round(x / n_s * 1e3, 3) for k, x in v._timing.items()})
if "profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        v.step(a)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
b.close()
