#!/usr/bin/env python3
"""Run under `rocprofv3 --pmc FETCH_SIZE` (and again with WRITE_SIZE): one calibration stream over the slot maps with
8-B and with 16-B loads (known byte count), then 1200 warm-up steps and 20 measured steps of the cfg2 bench loop.
tools/pmc_summary.py + the calibration factors turn the counters into HBM bytes per launch (profiles/*traffic*)."""
import os
import sys

os.environ["ORL_STREAMS"] = "1"  # whole-batch launches on one stream: the configuration the roofline pass of bench.py times

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS  # noqa: E402

fam, topo, kw, policy = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
B = 65536
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
env.run(policy, 1200)
for w in (0, 1, 0, 1):
    n = env.lib.orl_batch_debug_stream_read(env._h, w)
print("calibration bytes", n)
for _ in range(6):  # six launches of the persistent kernel, 20 steps each (tools/collect_profiles.py: STEPS)
    env.run(policy, 20)
for _ in range(20):  # the stand-alone slot-scan kernel (orl_batch_policy), on the same steady-state slot maps
    env.policy(policy)
env.close()
