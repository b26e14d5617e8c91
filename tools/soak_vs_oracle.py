#!/usr/bin/env python3
"""Soak: a long device-resident run of a BASELINE workload against the OpenMP oracle on EVERY env (counters, pending service,
slot maps, link and network statistics).  usage: soak_vs_oracle.py <workload> <batch> <steps> [episode_length]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS  # noqa: E402
from oracle.oracle import OracleBatch  # noqa: E402

wl, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fam, topo, kw, policy = WORKLOADS[wl]
kw = dict(kw, episode_length=int(sys.argv[4]) if len(sys.argv) > 4 else 1000)
seeds = [1000 + 7 * i for i in range(B)]
dev = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, **kw)
ora = OracleBatch(fam, topo, seeds, omp=True, **kw)
t0 = time.time()
done = 0
bad = 0
for chunk in (steps // 3, steps // 3, steps - 2 * (steps // 3)):  # three runs: the state carries over run boundaries
    dev.run(policy, chunk)
    ora.run(policy, chunk)
    done += chunk
    for name, a, b in (("counters", dev.counters(), ora.counters()), ("services", dev.services(), ora.services()),
                       ("active", dev.active(), ora.active()), ("slots", dev.slots_packed(), ora.slots_packed()),
                       ("link_stats", dev.link_stats_all(), ora.link_stats_all()), ("net_stats", dev.net_stats_all(), ora.net_stats_all())):
        a, b = np.asarray(a), np.asarray(b)
        eq = np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)
        if not eq:
            bad += 1
            print("MISMATCH", wl, name, "after", done, "steps: envs", np.unique(np.nonzero(a != b)[0])[:10])
print(wl, B, steps, "flags", int(dev.flags().any()), "mismatches", bad, "%.1f s" % (time.time() - t0))
dev.close()
sys.exit(1 if bad else 0)
