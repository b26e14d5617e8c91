#!/usr/bin/env python3
"""A host-side agent on the SB3-shaped VecEnv, overlapping its own bookkeeping with the device's step.

`OpticalVecEnv.step_async(actions)` queues the whole step on the batch's stream — the agent's action array as it is, the step
kernel, reward / done / observation back into page-locked arrays — and returns; `step_wait()` collects it.  What the agent does in
between (here: a running return estimate and an action-count table, standing in for a replay-buffer insert) costs no wall time
as long as it is shorter than the device's share.  The loop is the one of the reference's SB3 notebook
(examples/stable_baselines3/DeepRMSA.ipynb: Monitor -> DummyVecEnv -> PPO.learn), with a random policy in place of PPO.

    python examples/vec_env_host_agent.py [num_envs] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

import numpy as np  # noqa: E402

import optical_rl_gym_amd as orl  # noqa: E402
from optical_rl_gym_amd.vec_env import OpticalVecEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
batch = orl.make("DeepRMSA-v0", topology="nsfnet_chen", num_envs=n, seeds=7, mean_service_holding_time=7.5,
                 mean_service_inter_arrival_time=1.0 / 12.0, j=1, episode_length=1000)
venv = OpticalVecEnv(batch, obs_dtype=np.float32)
rng = np.random.default_rng(0)
obs = venv.reset()
n_actions = venv.action_space.n
counts = np.zeros(n_actions, np.int64)
returns = np.zeros(n)


def bookkeeping(prev_obs, prev_actions, prev_reward):
    """Stand-in for what an agent does with the PREVIOUS transition while the next step runs on the device."""
    counts[:] += np.bincount(prev_actions, minlength=n_actions)
    returns[:] = 0.99 * returns + prev_reward
    return float(prev_obs[:, 0].mean())


pool = [rng.integers(0, n_actions, n) for _ in range(8)]  # (drawn ahead: the generator is not what is being timed)
for overlap in (False, True):
    actions = pool[0]
    reward = np.zeros(n)
    t0 = time.perf_counter()
    for _ in range(steps):
        venv.step_async(actions)
        if overlap:
            bookkeeping(obs, actions, reward)                      # runs while the device steps
        new_obs, new_reward, done, infos = venv.step_wait()
        if not overlap:
            bookkeeping(obs, actions, reward)                      # runs after the device has finished
        obs, reward = new_obs, new_reward
        actions = pool[_ % 8]
    dt = time.perf_counter() - t0
    print("%s: %.3f ms per step, %.2e env-steps/s" % ("overlapped" if overlap else "one after the other", dt / steps * 1e3, n * steps / dt))
if venv.episode_log:  # (episodes last 999 steps here)
    print("episodes finished:", len(venv.episode_log), " mean episode_service_blocking_rate: %.4f"
          % np.mean([row["episode_service_blocking_rate"] for row in venv.episode_log]))
print("action counts:", counts.tolist())
venv.close()
