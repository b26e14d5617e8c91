"""ctypes front-end of the CPU oracle (oracle/orl_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  It deliberately does not import optical_rl_gym_amd: the
constructor-argument handling below restates the reference's __init__ chains on its own
(optical_network_env.py:14-94, rmsa_env.py:29-161, deeprmsa_env.py:10-46, rwa_env.py:19-94,
rmcsa_env.py:29-207) so the oracle and the product cross-check each other.
"""
import ctypes as C
import os
import random
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ENV_TYPES = {"RMSA": 0, "DeepRMSA": 1, "RWA": 2, "RMCSA": 3, "QoSConstrainedRA": 4}
POLICIES = {"SP_FF": 0, "SAP_FF": 1, "LLP_FF": 2, "SAP_LF": 3, "SP": 0, "SAP": 1, "SAP_BM_FC_FF": 1, "PATH_FF": 4}


class _Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "env_type", "n_nodes", "n_links", "k_paths", "max_hops", "n_mods", "num_slots", "num_cores",
        "episode_length", "allow_rejection", "j", "bit_rate_mode", "br_lo", "br_hi", "n_bit_rates", "n_classes")] + [
        (n, C.c_double) for n in ("mean_iat", "mean_ht", "channel_width", "worst_xt")]


class _Tables(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "n_paths", "path_hops", "path_links", "path_length", "path_best_mod", "mod_se", "mod_max_length",
        "mod_min_osnr", "mod_inband_xt", "edge_iter_order", "node_probs", "bit_rates", "bit_rate_probs", "class_probs",
        "class_reward")]


def build(force=False):
    so = os.path.join(HERE, "_build", "liborloracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "orl_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s"], stdout=subprocess.DEVNULL)
    return so


_LIBS = {}


def _lib(omp=False):
    key = bool(omp)
    if key not in _LIBS:
        so = build()
        if os.environ.get("ORL_ORACLE_SO") and not omp:  # e.g. the -fsanitize build of tools/sanitize_oracle.sh
            so = os.environ["ORL_ORACLE_SO"]
        if omp:
            so = so.replace("liborloracle.so", "liborloracle_omp.so")
        lib = C.CDLL(so)
        lib.orc_create.restype = C.c_void_p
        lib.orc_create.argtypes = [C.POINTER(_Config), C.POINTER(_Tables), C.c_int64, C.c_void_p]
        lib.orc_destroy.argtypes = [C.c_void_p]
        lib.orc_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_policy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_step.restype = C.c_int
        lib.orc_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_observation.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_run.restype = C.c_int64
        lib.orc_run.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        lib.orc_info_dim.argtypes = [C.POINTER(_Config)]
        lib.orc_obs_dim.argtypes = [C.POINTER(_Config)]
        for name in ("orc_get_service", "orc_get_counters"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p]
        for name in ("orc_get_slots", "orc_get_link_stats", "orc_get_net_stats"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.orc_get_spectrum.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.orc_set_paths.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_reseed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_get_action_histograms.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.orc_get_active.restype = C.c_int32
        lib.orc_get_active.argtypes = [C.c_void_p, C.c_int64]
        _LIBS[key] = lib
    return _LIBS[key]


def mt_states(seeds):
    """random.Random(seed).getstate() for every env: [n][625] uint32 (624 words + index)."""
    out = np.empty((len(seeds), 625), np.uint32)
    for i, s in enumerate(seeds):
        out[i] = random.Random(41 if s is None else int(s)).getstate()[1]
    return out


def load_topology_tables(name_or_path):
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(os.path.dirname(HERE), "optical_rl_gym_amd", "data", name_or_path + "_5-paths_6-modulations.npz")
    z = np.load(path)
    return {k: z[k] for k in z.files}


class OracleBatch:
    """n independent reference-equivalent envs, each constructed with seed=seeds[i]."""

    def __init__(self, env_type, topology, seeds, omp=False, **kw):
        self.lib = _lib(omp)
        t = load_topology_tables(topology) if isinstance(topology, str) else topology
        self.tables = t
        et = ENV_TYPES[env_type]
        N, E = len(t["node_names"]), len(t["link_length"])
        k, H, M = int(t["k_paths"]), t["path_links"].shape[-1], len(t["mod_se"])
        # ---- constructor defaults of the reference ----
        episode_length = kw.pop("episode_length", 1000)
        mht = kw.pop("mean_service_holding_time", 25.0 if et == 1 else 10800.0)
        if et == 1:  # deeprmsa_env.py:22-32
            miat_arg = kw.pop("mean_service_inter_arrival_time", 0.1)
            load = mht / miat_arg
            kw.pop("load", None)
        else:
            load = kw.pop("load", 10)
        miat = 1 / float(load / float(mht))  # optical_network_env.py:92-94
        S = kw.pop("num_spectrum_resources", 80 if et in (2, 4) else 100)
        allow_rejection = kw.pop("allow_rejection", True if et in (2, 4) else False)
        n_classes = int(kw.pop("num_service_classes", 1))  # qos_constrained_ra.py:21-23
        class_probs = np.asarray(kw.pop("classes_arrival_probabilities", [1.0]), np.float64)
        class_reward = np.asarray(kw.pop("classes_reward", [1.0]), np.float64)
        kw.pop("k_paths", None)
        probs = kw.pop("node_request_probabilities", None)
        probs = np.full(N, 1.0 / N) if probs is None else np.asarray(probs, np.float64)
        channel_width = kw.pop("channel_width", 50.0 if et == 2 else 12.5)
        mode = kw.pop("bit_rate_selection", "continuous")
        bit_rates = list(kw.pop("bit_rates", [10, 40, 100]))
        brp = kw.pop("bit_rate_probabilities", None)
        if brp is None:
            brp = [1.0 / len(bit_rates) for _ in range(len(bit_rates))]
        lo = int(kw.pop("bit_rate_lower_bound", 25))
        hi = int(kw.pop("bit_rate_higher_bound", 100))
        j = kw.pop("j", 1)
        cores = kw.pop("num_spatial_resources", 7) if et == 3 else 1
        inband = np.array(t["mod_inband_xt"], np.float64)
        worst_xt = 0.0
        if et == 3:  # rmcsa_env.py:63-67, 119-129
            worst_xt = kw.pop("worst_xt", None)
            if worst_xt is None:
                worst_xt = {7: -84.7, 12: -61.9, 19: -54.8}.get(cores)
            inband = inband + 4
            worst_xt = worst_xt + 4
        kw.pop("reset", None)
        assert not kw, "unknown kwargs %r" % kw
        cfg = _Config(et, N, E, k, H, M, S, cores, episode_length, int(bool(allow_rejection)), j,
                      1 if mode == "discrete" else 0, lo, hi, len(bit_rates), n_classes,
                      miat, float(mht), channel_width, worst_xt)
        self.cfg = cfg
        keep = dict(
            n_paths=np.ascontiguousarray(t["n_paths"], np.int32),
            path_hops=np.ascontiguousarray(t["path_hops"], np.int32),
            path_links=np.ascontiguousarray(t["path_links"], np.int32),
            path_length=np.ascontiguousarray(t["path_length"], np.float64),
            path_best_mod=np.ascontiguousarray(t["path_best_mod"], np.int32),
            mod_se=np.ascontiguousarray(t["mod_se"], np.int32),
            mod_max_length=np.ascontiguousarray(t["mod_max_length"], np.float64),
            mod_min_osnr=np.ascontiguousarray(t["mod_min_osnr"], np.float64),
            mod_inband_xt=np.ascontiguousarray(inband, np.float64),
            edge_iter_order=np.ascontiguousarray(t["edge_iter_order"], np.int32),
            node_probs=np.ascontiguousarray(probs, np.float64),
            bit_rates=np.array(bit_rates, np.int32),
            bit_rate_probs=np.array(brp, np.float64),
            class_probs=np.ascontiguousarray(class_probs), class_reward=np.ascontiguousarray(class_reward),
        )
        self._keep = keep
        tb = _Tables(*[keep[n].ctypes.data for n, _ in _Tables._fields_])
        self.n = len(seeds)
        self.n_info = self.lib.orc_info_dim(C.byref(cfg))
        self.obs_dim = self.lib.orc_obs_dim(C.byref(cfg))
        st = mt_states(seeds)
        self.h = self.lib.orc_create(C.byref(cfg), C.byref(tb), self.n, st.ctypes.data)
        self.C, self.E, self.S, self.k = cores, E, S, k

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_destroy(self.h)
            self.h = None

    def reset(self, full=False, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.orc_reset(self.h, int(full), None if m is None else m.ctypes.data)

    def set_paths(self, paths):
        p = np.ascontiguousarray(np.asarray(paths).reshape(self.n), np.int32)
        self.lib.orc_set_paths(self.h, p.ctypes.data)

    def seed(self, seeds, mask=None):
        """env.seed(seed) of the selected envs (optical_network_env.py:205-210)."""
        if np.isscalar(seeds):
            seeds = [int(seeds) + i for i in range(self.n)]
        st = mt_states(seeds)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.orc_reseed(self.h, st.ctypes.data, None if m is None else m.ctypes.data)

    # ---- bulk read-backs (every-env comparisons at full batch size) ----
    def slots_packed(self):
        """[n, cores*links*words (padded to even)] uint64, the product's bit-packed layout (bit s of a row = slot s free)."""
        words = (self.S + 63) // 64
        if self.S > 320:
            words = 8
        elif self.S > 128:
            words = 5
        elif self.S > 64:
            words = 2
        stride = (self.C * self.E * words + 1) & ~1
        out = np.zeros((self.n, stride), np.uint64)
        self.lib.orc_get_slots_packed_all.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64]
        self.lib.orc_get_slots_packed_all(self.h, out.ctypes.data, words, stride)
        return out

    def link_stats_all(self):
        out = np.zeros((self.n, 4, self.E))
        self.lib.orc_get_link_stats_all.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.orc_get_link_stats_all(self.h, out.ctypes.data)
        return out

    def net_stats_all(self):
        out = np.zeros((self.n, 4))
        self.lib.orc_get_net_stats_all.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.orc_get_net_stats_all(self.h, out.ctypes.data)
        return out

    def active(self):
        out = np.zeros(self.n, np.int32)
        self.lib.orc_get_active_all.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.orc_get_active_all(self.h, out.ctypes.data)
        return out

    def action_histograms_of(self, env=0):
        shape = (2, self.k + 1, self.S + 1)
        if self.cfg.env_type == 3:  # RMCSA (rmcsa_env.py:145-180)
            shape = (2, self.k + 1, self.cfg.n_mods + 1, self.C + 1, self.S + 1)
        out = np.zeros(shape, np.int64)
        self.lib.orc_get_action_histograms(self.h, env, out.ctypes.data)
        return out[0], out[1]

    def matrix_observation(self):
        """SimpleMatrixObservation (rmsa_env.py:806-837, rmcsa_env.py:914-947) from the read-back state."""
        N = len(self.tables["node_names"])
        svc = self.services()
        out = np.zeros((self.n, 2 * N + self.C * self.E * self.S), np.uint8)
        for i in range(self.n):
            src, dst = int(svc[i, 2]), int(svc[i, 3])
            out[i, min(src, dst)] = 1
            out[i, N + max(src, dst)] = 1
            out[i, 2 * N:] = self.slots(i).reshape(-1)
        return out

    def policy(self, policy, paths=None):
        if paths is not None:
            self.set_paths(paths)
        a = np.zeros((self.n, 4), np.int32)
        self.lib.orc_policy(self.h, POLICIES[policy] if isinstance(policy, str) else policy, a.ctypes.data)
        return a

    def step(self, actions, auto_reset=False):
        a = np.zeros((self.n, 4), np.int32)
        actions = np.asarray(actions)
        if actions.ndim == 1:
            actions = actions[:, None]
        a[:, : actions.shape[1]] = actions
        reward = np.zeros(self.n)
        done = np.zeros(self.n, np.uint8)
        info = np.zeros((self.n, self.n_info))
        obs = np.zeros((self.n, self.obs_dim)) if self.obs_dim else None
        rc = self.lib.orc_step(self.h, a.ctypes.data, int(auto_reset), reward.ctypes.data, done.ctypes.data,
                               info.ctypes.data, None if obs is None else obs.ctypes.data)
        if rc:
            raise IndexError("oracle step failed with code %d (the reference would raise here)" % rc)
        return obs, reward, done, info

    def run(self, policy, n_steps):
        return self.lib.orc_run(self.h, POLICIES[policy] if isinstance(policy, str) else policy, n_steps)

    def observation(self):
        obs = np.zeros((self.n, self.obs_dim))
        self.lib.orc_observation(self.h, obs.ctypes.data)
        return obs

    def services(self):
        out = np.zeros((self.n, 6))
        self.lib.orc_get_service(self.h, out.ctypes.data)
        return out

    def counters(self):
        out = np.zeros((self.n, 8), np.int64)
        self.lib.orc_get_counters(self.h, out.ctypes.data)
        return out

    def slots(self, env=0):
        out = np.zeros((self.C, self.E, self.S), np.uint8)
        self.lib.orc_get_slots(self.h, env, out.ctypes.data)
        return out

    def spectrum(self, env=0):
        """QoSConstrainedRA: topology.graph["available_spectrum"] (free units per link)."""
        out = np.zeros(self.E, np.int32)
        self.lib.orc_get_spectrum(self.h, env, out.ctypes.data)
        return out

    def link_stats(self, env=0):
        out = np.zeros((4, self.E))
        self.lib.orc_get_link_stats(self.h, env, out.ctypes.data)
        return out

    def net_stats(self, env=0):
        out = np.zeros(4)
        self.lib.orc_get_net_stats(self.h, env, out.ctypes.data)
        return out

    def n_active(self, env=0):
        return self.lib.orc_get_active(self.h, env)
