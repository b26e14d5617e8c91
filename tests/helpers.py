"""Shared helpers for the parity tests: golden-trace loading and trace replay."""
import glob
import json
import os
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    g["meta"] = json.loads(str(g["meta"]))
    return g


def crc_slots(slots_u8):
    """CRC32 of the dense 0/1 availability array, flattened [core][link][slot] (cores omitted when 1)."""
    return zlib.crc32(np.ascontiguousarray(slots_u8, np.uint8).tobytes())


def replay(env, g, check, n_steps=None):
    """Replay golden trace `g` on a 1-env batch object `env` (oracle or product; same method names).

    check(t, what, got, expected) is called for every compared quantity.
    """
    meta = g["meta"]
    T = meta["n_steps"] if n_steps is None else min(n_steps, meta["n_steps"])
    use_policy = meta["policy"] != "ACTIONS"
    for t in range(T):
        if g["reset_before"][t]:
            env.reset(full=False)
        check(t, "svc", env.services()[0], g["svc"][t])
        if "obs" in g:
            check(t, "obs", env.observation()[0], g["obs"][t])
        if use_policy:
            a = env.policy(meta["policy"])
            width = g["actions"].shape[1]
            check(t, "action", np.asarray(a[0, :width], np.int64), g["actions"][t])
        else:
            a = g["actions"][t][None, :]
        _, reward, done, info = env.step(a)
        check(t, "reward", reward[0], g["reward"][t])
        check(t, "done", int(done[0]), int(g["done"][t]))
        check(t, "info", info[0, : g["info"].shape[1]], g["info"][t])
        check(t, "counters", env.counters()[0], g["counters"][t])
        sl = env.slots(0)
        if sl.shape[0] == 1:
            sl = sl[0]
        check(t, "crc", crc_slots(sl), int(g["crc"][t]))
        check(t, "n_active", env.n_active(0), int(g["n_active"][t]))
        if (t + 1) in meta["snapshot_steps"]:
            s = t + 1
            packed = np.packbits(sl, axis=-1, bitorder="little")
            check(t, "snap_slots", packed, g["snap%d_slots" % s])
            check(t, "snap_link_stats", env.link_stats(0), g["snap%d_link_stats" % s])
            check(t, "snap_net_stats", env.net_stats(0), g["snap%d_net_stats" % s])
            if "snap%d_path_action_probability" % s in g:  # RWA vector infos ride behind the 2 scalars
                pa = g["snap%d_path_action_probability" % s]
                wa = g["snap%d_wavelength_action_probability" % s]
                check(t, "path_action_probability", info[0, 2 : 2 + len(pa)], pa)
                check(t, "wavelength_action_probability", info[0, 2 + len(pa) : 2 + len(pa) + len(wa)], wa)
    if T == meta["n_steps"]:
        check(T, "svc", env.services()[0], g["svc"][T])
