#!/usr/bin/env python3
"""Where k_rowstats' cycles go (diagnostic build: ORL_HIPCC_EXTRA=-DORL_RS_PROF, variant exp): cycles per phase summed over wavefronts.
    ORL_HIPCC_EXTRA=-DORL_RS_PROF ORL_LIB_VARIANT=exp tools/rs_prof.py [steps per launch] [launches]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ORL_PERSIST_VARIANT", "7")
import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fam, topo, kw, policy = WORKLOADS["cfg2"]
B = 65536
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
env.run(policy, 1500)
out = (C.c_ulonglong * 16)()
fn = env.lib.orl_debug_rs_prof  # (the CDLL handle: a symbol of the diagnostic build only)
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
fn(out, 1)
for _ in range(launches):
    env.run(policy, steps)
fn(out, 0)
v = [int(x) for x in out]
waves = v[10] or 1
names = ["setup", "A events", "A barrier", "B sort+row", "C touches", "C barrier", "D scan", "D barrier"]
tot = sum(v[:8])
for n, c in zip(names, v[:8]):
    print("%-12s %8.0f cycles per wavefront-window  %5.1f %%" % (n, c / waves, 100.0 * c / tot))
print("rounds per wavefront-window: max %.2f, mean per lane %.2f (lanes busy %.0f %%)" % (v[8] / waves, v[9] / waves / 64, 100.0 * v[9] / 64 / max(v[8], 1)))
env.close()
