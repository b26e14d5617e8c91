#!/usr/bin/env python3
"""Rows-deferred form of the persistent kernel (forms 7 / 8) against the library's default form, on EVERY env of a batch: counters,
pending services, pending releases, whole slot maps, link and network statistics after several runs with host steps in between.

    tools/rd_check.py [workload] [batch] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
fam, topo, kw, policy = WORKLOADS[workload]
kw = dict(kw, episode_length=90)
seeds = [77 + 3 * i for i in range(batch)]
out = {}
for name, var in (("default", None), ("rd7", "7"), ("rd8", "8")):
    if var is None:
        os.environ.pop("ORL_PERSIST_VARIANT", None)
    else:
        os.environ["ORL_PERSIST_VARIANT"] = var
    os.environ["ORL_PERSIST_RW"] = "0"
    env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
    env.run(policy, steps // 3)
    form1 = int(env.lib.orl_batch_debug_persist_form(env._h))
    a = env.policy(policy)
    env.step(a, auto_reset=True)
    env.run(policy, steps // 3)
    env.run(policy, steps - 2 * (steps // 3))
    form = int(env.lib.orl_batch_debug_persist_form(env._h))
    out[name] = dict(counters=env.counters().copy(), services=env.services().copy(), active=env.active().copy(), flags=env.flags().copy(),
                     slots=env.slots_packed().copy(), link=env.link_stats_all().copy(), net=env.net_stats_all().copy())
    print(name, "form", form1, form, "specialised", getattr(env, "specialised", None), flush=True)
    env.close()
bad = 0
for name in ("rd7", "rd8"):
    for key, ref in out["default"].items():
        got = out[name][key]
        same = np.array_equal(got, ref, equal_nan=True)
        if not same:
            bad += 1
            diff = np.argwhere(np.asarray(got) != np.asarray(ref))
            print("MISMATCH", name, key, "first at", diff[:3].tolist(), "of", len(diff))
print("rd_check", workload, batch, steps, "OK" if bad == 0 else "FAILED %d" % bad)
sys.exit(1 if bad else 0)
