#!/usr/bin/env python3
"""Rate of the loop an agent on the same GPU drives (nothing crosses PCIe): policy(fetch=False) [stand-alone slot scan, in
place of the agent's network] + step(None, auto_reset=True, fetch=False) for N steps, one sync at the end; the same with the
scan fused into the step launch (policy_step), and with the info entries SB3 does not read left out (set_info_mode); and of the
host-driven step() over PCIe.  usage: agent_loop_rate.py [workload] [batch]   (ORL_AGENT_STEP=0 -> the one-wavefront-per-env kernel)"""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import numpy as np  # noqa: F401,E402
import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS, workload_load  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
fam, topo, kw, policy = WORKLOADS[name]
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
env.run(policy, max(1500, int(math.ceil(5 * workload_load(kw)))))
out = dict(workload=name, batch=B, agent_step=os.environ.get("ORL_AGENT_STEP", "default"))
# step_only re-issues ONE set of actions (most are rejected from the second step on: no provision to apply) — the floor of the
# step kernel, not an agent's loop; policy_and_step / policy_step_fused apply fresh actions every step
for label, mode in (("step_only", 0), ("policy_and_step", 1), ("policy_step_fused", 2), ("policy_step_fused_rates_only", 3)):
    env.set_info_mode(mode == 3)
    env.policy(policy, fetch=False)
    env.sync()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        if mode >= 2:  # orl_batch_policy_step: the scan as the step kernel's first phase, one launch
            env.policy_step(policy, auto_reset=True, fetch=False)
            continue
        if mode:
            env.policy(policy, fetch=False)
        env.step(None, auto_reset=True, fetch=False)
    env.sync()
    dt = time.perf_counter() - t0
    out[label] = dict(us_per_step=round(dt / n * 1e6, 2), env_steps_per_s=round(B * n / dt, 1))
env.set_info_mode(False)
acts = env.policy(policy).copy()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    env.step(acts, auto_reset=True)
dt = time.perf_counter() - t0
out["host_driven_pcie"] = dict(us_per_step=round(dt / n * 1e6, 2), env_steps_per_s=round(B * n / dt, 1))
print(json.dumps(out))
env.close()
