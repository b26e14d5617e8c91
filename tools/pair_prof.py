#!/usr/bin/env python3
"""Shader-clock profile of a SPECIALISED persistent kernel (the library built with -DORL_TIMING through ORL_SPEC_EXTRA): cycles
per wavefront-step in each segment; with the two-wavefront form the control wavefront's waits (14: for the rows, 15: for the
statistics) and the row wavefront's idle time (40), early row store (41) and statistics (36-39).

    python3 tools/pair_prof.py [workload] [batch] [steps]      (on the GPU box; ORL_PERSIST_RW=0/1 selects the form)
"""
import ctypes as C
import math
import os
import sys

os.environ["ORL_JIT_SPEC"] = "1"
os.environ["ORL_SPEC_EXTRA"] = (os.environ.get("ORL_SPEC_EXTRA", "") + " -DORL_TIMING=1").strip()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS, workload_load  # noqa: E402
from optical_rl_gym_amd import _build  # noqa: E402

NAMES = {0: "loop head", 1: "slot scan (policy_g)", 2: "record words, action decode", 3: "path record + validation",
         4: "accept: ev_push, provision masks", 5: "reward, clock", 7: "next service from the look-ahead, log words",
         8: "done, event record", 10: "release_soon", 11: "record stores, item list", 13: "row phase: tail",
         20: "rel: soon-list loads", 21: "rel: rebuild scan", 22: "rel: candidate info + path record", 23: "rel: release loop",
         24: "rel: next_rel", 35: "row: record + clocks + row load", 36: "row: masks", 37: "row: after-summary",
         38: "row: f64 statistics", 39: "row: sums + stores"}
NAMES.update({12: "item list + signal", 14: "PAIR control: waits for the tables to be read", 15: "PAIR control: waits for the statistics",
              40: "PAIR row wavefront: idle", 41: "PAIR row: rows read, masks taken back, signal", 42: "PAIR row: end of step",
              43: "PAIR row: service look-ahead for the control wavefront"})
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 256
fam, topo, kw, policy = WORKLOADS[name]
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
buf = C.create_string_buffer(1024)
assert env.lib.orl_batch_spec_flags(env._h, buf, 1024) > 0
spec = C.CDLL(_build.spec_path(buf.value.decode() + " " + os.environ["ORL_SPEC_EXTRA"]))
spec.orl_spec_prof.argtypes = [C.c_void_p, C.c_int]
env.run(policy, max(1500, int(math.ceil(5 * workload_load(kw)))))
out = np.zeros(48, np.uint64)
assert spec.orl_spec_prof(out.ctypes.data, 1) == 0
st = env.run(policy, steps)
assert spec.orl_spec_prof(out.ctypes.data, 1) == 0
assert int(env.lib.orl_batch_debug_persist_spec(env._h)) in (1, 2)
waves = (B + 7) // 8
per = out.astype(np.float64) / (waves * steps)
print("%s B=%d RW=%s: %d steps in %.2f ms (%.1f us/step, timing build); cycles per workgroup-step:" %
      (name, B, os.environ.get("ORL_PERSIST_RW", "auto"), steps, st.ms_total, st.ms_total * 1e3 / steps))
for k in range(48):
    if per[k] > 0:
        print("  %2d %-48s %9.0f" % (k, NAMES.get(int(k), "?"), per[k]))
print("  control wavefront (slots 0-31): %.0f   row phase / row wavefront (32-47): %.0f" % (per[:32].sum(), per[32:].sum()))
env.close()
