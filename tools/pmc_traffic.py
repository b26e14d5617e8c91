#!/usr/bin/env python3
"""Workload driver for the rocprofv3 counter passes (tools/run_profiles.sh): state preparation to the steady state, one
calibration stream over the slot maps with 8-B and with 16-B loads (known byte count), then MEASURED launches of the
persistent kernel — 10 x run(policy, S), i.e. ten single S-step launches (S = 128: the production chunk length; S = 20: the
shape of the driver's `bench.py --steps 20` blocks) — and 20 launches of the stand-alone slot-scan kernel on the same
steady-state slot maps.  tools/collect_profiles.py turns the counters of the LAST 10 k_persist dispatches into per-launch /
per-step figures (profiles/traffic_<workload>.json).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- python3 tools/pmc_traffic.py cfg2 65536 [steps per launch]
"""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS, workload_load  # noqa: E402

MEASURED_LAUNCHES = 10

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
STEPS_PER_LAUNCH = int(sys.argv[3]) if len(sys.argv) > 3 else 128
fam, topo, kw, policy = WORKLOADS[name]
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
def run(n_steps):
    try:
        env.run(policy, n_steps)
    except (OverflowError, IndexError):  # diagnostic builds that leave a phase out (-DORL_X_SKIP_*): the counters are what is wanted
        if not os.environ.get("ORL_X_IGNORE_ERRORS"):
            raise


run(max(1500, int(math.ceil(5 * workload_load(kw)))))
for w in (0, 1, 0, 1):
    n = env.lib.orl_batch_debug_stream_read(env._h, w)
for _ in range(MEASURED_LAUNCHES):
    run(STEPS_PER_LAUNCH)
for _ in range(20):
    env.policy(policy, fetch=False)
env.sync()
print(json.dumps(dict(workload=name, batch=B, calibration_bytes=int(n), steps_per_launch=STEPS_PER_LAUNCH,
                      measured_launches=MEASURED_LAUNCHES, mean_active_services=float(env.active().mean()))))
env.close()
