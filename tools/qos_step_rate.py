#!/usr/bin/env python3
"""QoSConstrainedRA with an agent in the loop: policy(fetch=False) + step(None, auto_reset=True, fetch=False) on the device, one
sync at the end — through the 8-lanes-per-env kernel (k_agent_qos, the library's choice from 20 480 envs) and through the
one-wavefront-per-env kernel (ORL_AGENT_STEP=0), on the same seeds; and the two must leave the same state.

    python3 tools/qos_step_rate.py [batch]        (on the GPU box)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kw = dict(load=300, mean_service_holding_time=25, episode_length=50, num_spectrum_resources=64, num_service_classes=3,
          classes_arrival_probabilities=[0.2, 0.5, 0.3], classes_reward=[4.0, 2.0, 1.0], allow_rejection=True)
out = {}
for label, knob in (("k_agent_qos (8 lanes per env)", "1"), ("k_step (one wavefront per env)", "0")):
    os.environ["ORL_AGENT_STEP"] = knob
    import optical_rl_gym_amd as orl

    env = orl.make("QoSConstrainedRA-v0", topology="nsfnet_chen", num_envs=B, seeds=[10 + i for i in range(B)], **kw)
    env.run("SAP_FF", 1500)
    res = {}
    for what in ("step_only", "policy_and_step"):
        env.policy("SAP_FF", fetch=False)
        env.sync()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            if what == "policy_and_step":
                env.policy("SAP_FF", fetch=False)
            env.step(None, auto_reset=True, fetch=False)
        env.sync()
        res[what] = (time.perf_counter() - t0) / n * 1e6
    best = min(env.run("SAP_FF", 200).ms_total for _ in range(3))
    res["run"] = best / 200 * 1e3
    out[knob] = (env.counters().copy(), env.services().copy(), env.link_stats_all().copy())
    print("%-34s %d envs: step only %.1f us (%.3e env-steps/s), scan + step %.1f us (%.3e), device loop run() %.1f us per step (%.3e)" %
          (label, B, res["step_only"], B / res["step_only"] * 1e6, res["policy_and_step"], B / res["policy_and_step"] * 1e6,
           res["run"], B / res["run"] * 1e6), flush=True)
    env.close()
print("state after the same 1 500 + 400 + 600 steps:", "equal" if all(np.array_equal(x, y, equal_nan=True) for x, y in zip(out["1"], out["0"])) else "DIFFERS")
