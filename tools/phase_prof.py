#!/usr/bin/env python3
"""Per-phase shader-clock profile of the persistent kernel (diagnostic build liborlgpu_timing.so, -DORL_TIMING):
cycles per wavefront-step in each segment of the control phase, the release detection and the row phase.

    python3 tools/phase_prof.py [workload] [batch] [steps]        (on the GPU box; ORL_PERSIST_VARIANT selects the kernel form)
"""
import math
import os
import sys

os.environ["ORL_LIB_VARIANT"] = "timing"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS, workload_load  # noqa: E402

NAMES = {0: "loop head", 1: "slot scan (policy_g)", 2: "env_load, pending g_comp, action decode", 3: "path record + validation",
         4: "accept: counters, ev_push, provision masks", 5: "histograms, reward, record stores", 6: "rng_fill + network stats",
         7: "next_service (MT, 2 x log, choices)", 8: "done / auto reset / env_store", 10: "release_soon: tail", 11: "release stores",
         9: "emit", 12: "item list + barriers", 13: "row phase: tail",
         20: "rel: soon-list loads", 21: "rel: rebuild scan", 22: "rel: candidate info + path record", 23: "rel: release loop",
         24: "rel: next_rel", 35: "row: record + clocks", 36: "row: load row, before-summary, masks", 37: "row: after-summary",
         38: "row: f64 statistics", 39: "row: stores"}

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 256
fam, topo, kw, policy = WORKLOADS[name]
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
env.run(policy, max(1500, int(math.ceil(5 * workload_load(kw)))))
out = np.zeros(48, np.uint64)
env._ck(env.lib.orl_batch_debug_prof(env._h, out.ctypes.data, 1))
st = env.run(policy, steps)
env._ck(env.lib.orl_batch_debug_prof(env._h, out.ctypes.data, 1))
waves = (B + 7) // 8
per = out.astype(np.float64) / (waves * steps)
print("%s B=%d: %d steps in %.2f ms (%.1f us/step, timing build); cycles per wavefront-step:" % (name, B, steps, st.ms_total, st.ms_total * 1e3 / steps))
tot = per.sum()
for k in np.argsort(-per):
    if per[k] > 0:
        print("  %2d %-48s %9.0f  %5.1f %%" % (k, NAMES.get(int(k), "?"), per[k], 100 * per[k] / tot))
print("  total %.0f cycles per wavefront-step" % tot)
env.close()
