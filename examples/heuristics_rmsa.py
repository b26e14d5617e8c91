#!/usr/bin/env python3
"""The reference's tests/test_rmsa.py flow on the MI355X build: same kwargs, same heuristics, same printed numbers
(SP-FF 88.7000 +- 7.1281, SAP-FF 95.0000 +- 3.2558, LLP-FF 95.1000 +- 3.3897), then the same policy on a 4 096-env batch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

import optical_rl_gym_amd as orl  # noqa: E402

env_args = dict(topology="nsfnet_chen", seed=10, allow_rejection=True, load=50, mean_service_holding_time=25,
                episode_length=100, num_spectrum_resources=64, bit_rate_selection="discrete")

print("STR".ljust(8), "REW".rjust(8), "STD".rjust(8))
for name, heuristic in (("SP-FF", orl.shortest_path_first_fit), ("SAP-FF", orl.shortest_available_path_first_fit),
                        ("LLP-FF", orl.least_loaded_path_first_fit)):
    env = orl.RMSAEnv(**env_args)
    mean_reward, std_reward = orl.evaluate_heuristic(env, heuristic, n_eval_episodes=10)
    print((name + ":").ljust(8), f"{mean_reward:.4f}  {std_reward:>7.4f}")
    print("\tBit rate blocking:", (env.episode_bit_rate_requested - env.episode_bit_rate_provisioned) / env.episode_bit_rate_requested)
    print("\tRequest blocking:", (env.episode_services_processed - env.episode_services_accepted) / env.episode_services_processed)
    env.close()

kw = dict(env_args)
kw.pop("seed")
batch = orl.make("RMSA-v0", num_envs=4096, seeds=10, **kw)  # seeds 10, 11, 12, ...
t0 = time.time()
batch.run("SAP_FF", 1000)
batch.sync()
dt = time.time() - t0
processed, accepted = batch.totals()
print("4096 envs x 1000 KSP-FF steps: %.2f s, %.1f M env-steps/s, blocking %.4f" % (dt, 4096 * 1000 / dt / 1e6, 1 - accepted / processed))
batch.close()
