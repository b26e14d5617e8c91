#!/usr/bin/env python3
"""Pre-build specialisation libraries with extra compiler flags (A/B experiments on the specialised kernels: the flags are
part of the cache key, ORL_SPEC_EXTRA selects them at run time).  Cross-compiles without a GPU.

    tools/ab_spec.py cfg2,cfg3 "" "-mllvm -disable-machine-licm" "-DORL_X_KARG -mllvm -disable-machine-licm"
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
from optical_rl_gym_amd import _build, envs  # noqa: E402

for name in sys.argv[1].split(","):
    fam, topo, kw, _ = WORKLOADS[name]
    flags = envs.ENV_CLASSES[fam].spec_flags(topology=topo, **kw)
    for extra in sys.argv[2:]:
        f = flags + (" " + extra if extra else "")
        print(name, repr(extra), os.path.basename(_build.build_spec(f)), flush=True)
