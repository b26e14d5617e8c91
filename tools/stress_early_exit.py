#!/usr/bin/env python3
"""The rare paths of the persistent kernel made common: with the item form limited to one or two releases per step
(ORL_ITEM_MASKS) wavefronts leave their launches early thousands of times — releases done in place at the next launch, services
drawn ahead parked and picked up again, in the two-wavefront form of small batches the batch on order taken over at the exit and
groups out of phase asking on the spot.  Every env of the device loop against the one-wavefront-per-env kernel.

    python3 tools/stress_early_exit.py [workload ...] [--envs N] [--steps K]        (on the GPU box)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=["cfg2", "cfg3", "cfg1", "cfg5"])
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--steps", type=int, default=700)
args = ap.parse_args()
os.environ["ORL_JIT_SPEC"] = "1"
import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS  # noqa: E402

bad = 0
for name in args.workloads:
    fam, topo, kw, policy = WORKLOADS[name]
    kw = dict(kw, episode_length=70)
    seeds = [5 + 11 * i for i in range(args.envs)]
    for masks in ("1", "2"):
        out = {}
        for impl in ("64", "2"):
            os.environ["ORL_STEP_IMPL"] = impl
            os.environ["ORL_PERSIST"] = "0" if impl == "64" else "1"
            if impl == "64":
                os.environ.pop("ORL_ITEM_MASKS", None)
            else:
                os.environ["ORL_ITEM_MASKS"] = masks
            env = orl.make(fam, topology=topo, num_envs=args.envs, seeds=seeds, **kw)
            for chunk in (args.steps // 3, 97, args.steps - args.steps // 3 - 97):
                env.run(policy, chunk)
            form = int(env.lib.orl_batch_debug_persist_spec(env._h))
            serial = int(env.lib.orl_batch_debug_serial_count(env._h))
            out[impl] = (env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(),
                         env.net_stats_all().copy(), env.link_stats_all().copy(), env.slots_packed().copy(), form, serial)
            env.close()
        ok = all(np.array_equal(out["64"][k], out["2"][k], equal_nan=True) if out["64"][k].dtype.kind == "f" else np.array_equal(out["64"][k], out["2"][k])
                 for k in range(7))
        bad += 0 if ok else 1
        print("%s %d envs x %d steps, item masks %s: kernel form %d (2 = pair), %d env-steps released in place: %s" %
              (name, args.envs, args.steps, masks, out["2"][7], out["2"][8], "every env equal" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
