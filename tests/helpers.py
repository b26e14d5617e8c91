"""Shared helpers for the parity tests: golden-trace loading and trace replay."""
import glob
import json
import os
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_names(prefix="g"):
    """g* = step traces (gen_golden.py), w* = wrappers / seed / full reset (gen_golden_wrappers.py)."""
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


def replay_w(env, g, check):
    """Replay a w* fixture (oracle/gen_golden_wrappers.py) on a 1-env batch object `env` (oracle or product):
    PathOnlyFirstFitAction through policy "PATH_FF", SimpleMatrixObservation through matrix_observation(), seed() and
    full resets at the recorded steps, the 2-D action histograms at the end."""
    meta = g["meta"]
    events = {int(k): v for k, v in meta.get("events", {}).items()}
    wrapper = meta.get("wrapper")
    done = True
    for t in range(meta["n_steps"]):
        for kind, arg in events.get(t, []):
            if kind == "seed":
                env.seed([arg])
            else:
                env.reset(full=True)
                done = False
        if done:
            env.reset(full=False)
        check(t, "svc", env.services()[0], g["svc"][t])
        if "obs_bits" in g:
            obs = np.asarray(env.matrix_observation()[0], np.uint8)
            check(t, "matrix_obs", np.packbits(obs, bitorder="little"), g["obs_bits"][t])
            assert obs.size == int(g["obs_dim"])
        if wrapper == "PathOnlyFirstFitAction":
            a = env.policy("PATH_FF", paths=[int(g["choice"][t][0])])
        else:
            a = env.policy(meta["policy"])
        width = g["actions"].shape[1]
        check(t, "action", np.asarray(a[0, :width], np.int64), g["actions"][t])
        _, reward, done_a, info = env.step(a)
        done = bool(done_a[0])
        check(t, "reward", reward[0], g["reward"][t])
        check(t, "done", int(done), int(g["done"][t]))
        check(t, "info", info[0, : g["info"].shape[1]], g["info"][t])
        check(t, "counters", env.counters()[0], g["counters"][t])
        sl = env.slots(0)
        if sl.shape[0] == 1:
            sl = sl[0]
        check(t, "crc", crc_slots(sl), int(g["crc"][t]))
        check(t, "n_active", env.n_active(0), int(g["n_active"][t]))
    T = meta["n_steps"]
    check(T, "svc", env.services()[0], g["svc"][T])
    sl = env.slots(0)
    if sl.shape[0] == 1:
        sl = sl[0]
    check(T, "final_slots", np.packbits(sl, axis=-1, bitorder="little"), g["final_slots"])
    check(T, "final_link_stats", env.link_stats(0), g["final_link_stats"])
    check(T, "final_net_stats", env.net_stats(0), g["final_net_stats"])
    if "actions_output" in g:
        out, taken = env.action_histograms_of(0)
        ro, rt = g["actions_output"], g["actions_taken"]
        check(T, "actions_output", out[: ro.shape[0], : ro.shape[1]], ro)
        check(T, "actions_taken", taken[: rt.shape[0], : rt.shape[1]], rt)
        assert out.sum() == ro.sum() and taken.sum() == rt.sum()


def replay_h(env, g, check):
    """Replay the h1 fixture (oracle/gen_golden_hist.py): RMCSA under a stored action stream with a full reset in the
    middle; the 4-D actions_output / actions_taken arrays (rmcsa_env.py:145-180) right before the reset and at the end,
    compared through their non-zero cells."""
    meta = g["meta"]

    def cells(a):
        flat = np.asarray(a, np.int64).ravel()
        idx = np.flatnonzero(flat)
        return np.stack([idx, flat[idx]], 1)

    done = True
    for t in range(meta["n_steps"]):
        if t == meta["reset_at"]:
            out, taken = env.action_histograms_of(0)
            assert list(out.shape) == meta["shape"]
            check(t, "actions_output before reset", cells(out), g["out_before"])
            check(t, "actions_taken before reset", cells(taken), g["taken_before"])
            env.reset(full=True)
            done = False
        if done:
            env.reset(full=False)
        _, reward, done_a, _ = env.step(g["actions"][t][None, :])
        done = bool(done_a[0])
        check(t, "reward", reward[0], g["reward"][t])
    out, taken = env.action_histograms_of(0)
    check(meta["n_steps"], "actions_output", cells(out), g["out_final"])
    check(meta["n_steps"], "actions_taken", cells(taken), g["taken_final"])
    check(meta["n_steps"], "counters", env.counters()[0], g["counters"])


def replay_q(env, g, check):
    """Replay a q* fixture (QoSConstrainedRA, oracle/gen_golden_qos.py) on a 1-env batch object `env`."""
    meta = g["meta"]
    use_policy = meta["policy"] != "ACTIONS"
    for t in range(meta["n_steps"]):
        if g["reset_before"][t]:
            env.reset(full=False)
        check(t, "svc", env.services()[0], g["svc"][t])
        if use_policy:
            a = env.policy(meta["policy"])
            check(t, "action", np.asarray(a[0, :1], np.int64), g["actions"][t])
        else:
            a = g["actions"][t][None, :]
        _, reward, done, info = env.step(a)
        check(t, "reward", reward[0], g["reward"][t])
        check(t, "done", int(done[0]), int(g["done"][t]))
        check(t, "info", info[0, :2], g["info"][t])
        check(t, "counters", env.counters()[0, :4], g["counters"][t][:4])
        check(t, "spectrum", env.spectrum(0), g["spectrum"][t])
        check(t, "n_active", env.n_active(0), int(g["n_active"][t]))
        if (t + 1) in meta["snapshot_steps"]:
            ls = env.link_stats(0)
            ref = g["snap%d_link_stats" % (t + 1)]
            check(t, "utilization", ls[0], ref[0])
            check(t, "last_update", ls[3], ref[3])
    check(meta["n_steps"], "svc", env.services()[0], g["svc"][meta["n_steps"]])
