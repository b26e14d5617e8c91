#!/usr/bin/env python3
"""RMCSA's 4-D action histograms (rmcsa_env.py:145-180, 219, 273, 284-289, 437-454) captured from the reference.

Runs in the BUILD container only (imports /root/reference through oracle/refshim, like gen_golden.py):
    python oracle/gen_golden_hist.py
Writes tests/golden/h1_rmcsa_hist4d.npz: a stored random action stream (in range, busy, out of range, reject), a full
reset in the middle (which clears both arrays), and the non-zero cells of actions_output / actions_taken at the end and
right before the reset, as (flat index, count) pairs — the dense arrays are 6 x 7 x 8 x 101 cells.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up the shim, numpy.int, the reference import)

import gym  # noqa: E402  (the stand-in under oracle/refshim)


def sparse(a):
    flat = np.asarray(a, np.int64).ravel()
    idx = np.flatnonzero(flat)
    return np.stack([idx, flat[idx]], 1).astype(np.int64)


def main():
    kw = dict(seed=31, allow_rejection=True, load=400, mean_service_holding_time=25, episode_length=250,
              num_spectrum_resources=100, num_spatial_resources=7)
    env = gym.make("RMCSA-v0", topology=gg.load_topology("nsfnet_chen"), **kw)
    n_steps, reset_at = 1400, 600
    acts = gg.random_actions(977, n_steps, 5, 100, extra=(6, 7))
    rewards, before = [], None
    done = True
    for t in range(n_steps):
        if t == reset_at:
            before = (sparse(env.actions_output), sparse(env.actions_taken))
            env.reset(only_episode_counters=False)
            done = False
        if done:
            env.reset()
        _, r, done, _ = env.step([int(x) for x in acts[t]])
        rewards.append(r)
    meta = dict(env="RMCSA", topology="nsfnet_chen", kwargs=kw, n_steps=n_steps, reset_at=reset_at,
                shape=list(env.actions_output.shape))
    np.savez_compressed(os.path.join(gg.GOLD, "h1_rmcsa_hist4d.npz"), meta=np.array(json.dumps(meta)),
                        actions=np.asarray(acts, np.int64), reward=np.asarray(rewards, np.float64),
                        out_before=before[0], taken_before=before[1],
                        out_final=sparse(env.actions_output), taken_final=sparse(env.actions_taken),
                        counters=np.asarray(gg.counters(env), np.int64))
    print("h1_rmcsa_hist4d: %d steps, accepted %d, cells %d / %d" % (n_steps, gg.counters(env)[1], len(sparse(env.actions_output)),
                                                                  len(sparse(env.actions_taken))))


if __name__ == "__main__":
    main()
