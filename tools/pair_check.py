#!/usr/bin/env python3
"""The two-wavefront form of the persistent kernel (ORL_PERSIST_RW) against the one-wavefront form: same state after the
same runs, and the rate of both.

    python3 tools/pair_check.py [workload ...] [--envs N] [--steps K]        (on the GPU box)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=["cfg2", "cfg3", "cfg1"])
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--warm", type=int, default=1500)
args = ap.parse_args()

import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS  # noqa: E402

os.environ["ORL_JIT_SPEC"] = "1"
for name in args.workloads:
    fam, topo, kw, policy = WORKLOADS[name]
    out = {}
    for rw in ("0", "1"):
        os.environ["ORL_PERSIST_RW"] = rw
        env = orl.make(fam, topology=topo, num_envs=args.envs, seeds=[10 + i for i in range(args.envs)], **kw)
        env.run(policy, args.warm)
        env.run(policy, 77)
        ran = int(env.lib.orl_batch_debug_persist_spec(env._h))
        if ran != (2 if rw == "1" else 1):
            print("   %s %d envs: ORL_PERSIST_RW=%s ran form %d (0 generic, 1 specialised, 2 pair): the pair needs the slot maps in LDS" % (name, args.envs, rw, ran))
        pick = (0, 1, args.envs // 2, args.envs - 1)
        state = [env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy()] + \
                [env.slots(i).copy() for i in pick] + [env.link_stats(i).copy() for i in pick] + [env.net_stats(i).copy() for i in pick]
        best = min(env.run(policy, args.steps).ms_total for _ in range(args.reps))
        out[rw] = (state, args.envs * args.steps / best * 1e3)
        env.close()
    names = ["counters", "services", "active", "flags"] + ["slots"] * 4 + ["link_stats"] * 4 + ["net_stats"] * 4
    same = True
    for nm, x, y in zip(names, out["0"][0], out["1"][0]):
        if not np.array_equal(x, y, equal_nan=True):
            same = False
            bad = np.argwhere(np.asarray(x) != np.asarray(y))
            print("   %s differs at %d places, first %s: %r vs %r" % (nm, len(bad), bad[0], np.asarray(x)[tuple(bad[0])], np.asarray(y)[tuple(bad[0])]))
    print("%s %d envs: one wavefront %.3e, pair %.3e env-steps/s (%+.1f %%), state %s" %
          (name, args.envs, out["0"][1], out["1"][1], 100 * (out["1"][1] / out["0"][1] - 1), "equal" if same else "DIFFERS"), flush=True)
