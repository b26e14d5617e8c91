"""OpticalVecEnv.step() rate at 65 536 envs (host-facing SB3 interface: actions in, numpy results + info dicts out).
usage (GPU box): python tools/vec_env_rate.py"""
import sys,time
import os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
import optical_rl_gym_amd as orl
from optical_rl_gym_amd.vec_env import OpticalVecEnv
from bench import WORKLOADS
for name in ("cfg3","cfg2"):
    fam,topo,kw,pol=WORKLOADS[name]
    B=65536
    b=orl.make(fam, topology=topo, num_envs=B, seeds=[10+i for i in range(B)], **kw)
    v=OpticalVecEnv(b, obs_dtype=np.float32 if "f32" in sys.argv else np.float64)
    v.reset()
    a=b.policy(pol)[:, :b.N_ACTION].copy() if fam!="DeepRMSA" else b.policy(pol)[:,0].copy()
    v.step(a)
    t0=time.perf_counter(); n=20
    for _ in range(n): v.step(a)
    dt=time.perf_counter()-t0
    print(name, "VecEnv.step: %.2f ms per step, %.3e env-steps/s"%(dt/n*1e3, B*n/dt))
    b.close()
