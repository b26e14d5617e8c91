#!/usr/bin/env python3
"""Fixed cost of one orl_batch_run call: tiny batch (the kernels take no time), n_steps 0 / 1 / 2, plus a 3 072-wavefront batch
at 1..3 steps (one generation: per-wavefront prologue + first steps)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import optical_rl_gym_amd as orl  # noqa: E402

kw = dict(load=300, mean_service_holding_time=25, episode_length=1000, num_spectrum_resources=320)
for B in (64, 24576, 65536):
    env = orl.make("RMSA", topology="nsfnet_chen", num_envs=B, seeds=[10 + i for i in range(B)], **kw)
    env.run("SAP_FF", 1500)
    for n in (0, 1, 2, 3, 4, 8, 20):
        ts = []
        for _ in range(30):
            env.sync()
            t0 = time.perf_counter()
            st = env.run("SAP_FF", n)
            env.sync()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("B=%6d n_steps=%2d  wall median %7.1f us  min %7.1f us   device (events) %7.1f us" % (B, n, ts[15] * 1e6, ts[0] * 1e6, st.ms_total * 1e3))
    env.close()
