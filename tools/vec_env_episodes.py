"""OpticalVecEnv.step() at the steady state of cfg3 (DeepRMSA, 50-step episodes, 65 536 envs): ~1 300 envs finish an episode per step,
each of which SB3's VecEnv contract gives an info dict of its own.  usage (GPU box): python tools/vec_env_episodes.py"""
import sys,os,time
sys.path.insert(0,'/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np, torch
import optical_rl_gym_amd as orl
from optical_rl_gym_amd.vec_env import OpticalVecEnv
from bench import WORKLOADS
fam,topo,kw,pol=WORKLOADS["cfg3"]
B=65536
b=orl.make(fam, topology=topo, num_envs=B, seeds=[10+i for i in range(B)], **kw)
v=OpticalVecEnv(b, obs_dtype=np.float32)
v.reset()
a=b.policy(pol)[:,0].copy()
for _ in range(60): v.step(a)
t0=time.perf_counter(); n=100; fin=0
for _ in range(n):
    o,r,d,i=v.step(a); fin+=int(d.sum())
dt=time.perf_counter()-t0
print("cfg3 (episode_length 50) VecEnv.step f32: %.2f ms per step, %d envs finish an episode per step"%(dt/n*1e3, fin/n))
