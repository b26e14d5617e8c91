"""Batched optical-network environments on MI355X: host-side mirror of the reference's env classes.

Each class takes the reference constructor's kwargs (same names, same defaults, same derivations) plus
`num_envs`, `seeds` and `device_id`, flattens them into the tables of include/orl.h and drives the HIP
library through ctypes.  Env i of a batch behaves exactly like a reference env built with seed=seeds[i].

reference constructors: optical_network_env.py:14-94, rmsa_env.py:29-161, deeprmsa_env.py:10-46,
rwa_env.py:19-94, rmcsa_env.py:29-207.
"""
import ctypes as C
import itertools
import math
import os
import random

import numpy as np

from . import _build, _lib
from .topology import Topology

POLICIES = {"SP_FF": 0, "SAP_FF": 1, "KSP_FF": 1, "LLP_FF": 2, "SAP_LF": 3, "SP": 0, "SAP": 1, "SAP_BM_FC_FF": 1,
            "PATH_FF": 4}

RMSA_INFO_KEYS = ["service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate",
                  "episode_bit_rate_blocking_rate", "network_compactness", "network_compactness_difference",
                  "avg_link_compactness", "avg_link_utilization"]
COUNTER_NAMES = ["services_processed", "services_accepted", "episode_services_processed",
                 "episode_services_accepted", "bit_rate_requested", "bit_rate_provisioned",
                 "episode_bit_rate_requested", "episode_bit_rate_provisioned"]


def mt_states(seeds):
    """optical_network_env.py:205-210: rng = random.Random(seed or 41) -> [n][625] uint32 MT19937 state."""
    out = np.empty((len(seeds), 625), np.uint32)
    for i, s in enumerate(seeds):
        out[i] = random.Random(41 if s is None else int(s)).getstate()[1]
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data



class _PinnedBlock:
    """Owner of one page-locked host allocation (orl_host_alloc); numpy views keep it alive through their base chain."""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.orl_host_free(self.ptr)
        except Exception:
            pass


class BatchedOpticalEnv:
    """Common machinery; use one of the four family classes below."""

    ENV_TYPE = None
    N_ACTION = 2

    # ---- construction --------------------------------------------------------------------------
    def _setup(self, topology, num_envs, seeds, device_id, *, episode_length, load, mean_service_holding_time,
               num_spectrum_resources, allow_rejection, node_request_probabilities, channel_width,
               bit_rate_selection="continuous", bit_rates=(10, 40, 100), bit_rate_probabilities=None,
               bit_rate_lower_bound=25, bit_rate_higher_bound=100, j=1, num_spatial_resources=1,
               modulations=None, worst_xt=None, event_capacity=0, action_histograms=False, num_service_classes=1,
               classes_arrival_probabilities=(1.0,), classes_reward=(1.0,)):
        self.lib = _lib.lib()  # ORL_LIB_VARIANT=alt selects the -DORL_ALT_IMPLS build (cross-implementation tests)
        self.action_histograms = bool(action_histograms)
        self.topology = Topology.load(topology) if isinstance(topology, str) else topology
        t = self.topology
        self.num_envs = int(num_envs)
        if seeds is None:
            seeds = [None] * self.num_envs
        elif np.isscalar(seeds):
            seeds = [int(seeds) + i for i in range(self.num_envs)]
        assert len(seeds) == self.num_envs
        self.seeds = list(seeds)
        self.device_id = device_id
        self.episode_length = episode_length
        self.allow_rejection = bool(allow_rejection)
        self.reject_action = 1 if allow_rejection else 0
        self.num_spectrum_resources = num_spectrum_resources
        self.num_spatial_resources = num_spatial_resources
        self.k_paths = t.k_paths
        self.channel_width = channel_width
        self.j = j
        # set_load (optical_network_env.py:76-94)
        self.load = load
        self.mean_service_holding_time = mean_service_holding_time
        self.mean_service_inter_arrival_time = 1 / float(load / float(mean_service_holding_time))
        lambda_a = 1 / self.mean_service_inter_arrival_time  # rmsa_env.py:548-550
        lambda_h = 1 / self.mean_service_holding_time        # rmsa_env.py:553
        # node pair tables (optical_network_env.py:68-74, 156-173)
        N = t.n_nodes
        if node_request_probabilities is None:
            probs = np.full(N, fill_value=1.0 / N)
        else:
            probs = np.asarray(node_request_probabilities, dtype=np.float64)
            assert len(probs) == N
        self.node_request_probabilities = probs
        cum_src = np.array(list(itertools.accumulate(probs)), np.float64)
        cum_dst = np.zeros((N, N), np.float64)
        for s in range(N):
            w = np.copy(probs)
            w[s] = 0.0
            w = w / np.sum(w)
            cum_dst[s] = list(itertools.accumulate(w))
        # bit rates (rmsa_env.py:78-99)
        self.bit_rate_selection = bit_rate_selection
        mods = list(t.modulations if modulations is None else modulations)
        self.modulation_formats = mods
        M = len(mods)
        if self.ENV_TYPE in (2, 4):
            table_rates, cum_br, mode = [0], None, 0
            lo = hi = 0
        elif bit_rate_selection == "continuous":
            lo, hi = int(bit_rate_lower_bound), int(bit_rate_higher_bound)
            assert lo == bit_rate_lower_bound and hi == bit_rate_higher_bound
            table_rates, cum_br, mode = list(range(lo, hi + 1)), None, 0
        else:
            assert bit_rate_selection == "discrete"
            if bit_rate_probabilities is None:
                bit_rate_probabilities = [1.0 / len(bit_rates) for _ in range(len(bit_rates))]
            assert len(bit_rates) == len(bit_rate_probabilities)
            self.bit_rates = list(bit_rates)
            self.bit_rate_probabilities = list(bit_rate_probabilities)
            table_rates = [int(b) for b in bit_rates]
            cum_br = np.array(list(itertools.accumulate(bit_rate_probabilities)), np.float64)
            mode, lo, hi = 1, 0, 0
        # get_number_slots (rmsa_env.py:610-621): ceil(bit_rate / (se * channel_width)) + 1
        n_slots = np.zeros((len(table_rates), M), np.uint8)
        if self.ENV_TYPE not in (2, 4):
            for i, br in enumerate(table_rates):
                for m, mod in enumerate(mods):
                    n_slots[i, m] = math.ceil(br / (mod.spectral_efficiency * channel_width)) + 1
        # RMCSA reach tables (_crosstalk_is_acceptable, rmcsa_env.py:341-384), evaluated with the reference's
        # own float expressions for every (modulation, bit rate)
        lmax_snr = lmax_xt = None
        if self.ENV_TYPE == 3:
            self.worst_xt = worst_xt
            lmax_snr = np.zeros((M, len(table_rates)), np.float64)
            lmax_xt = np.zeros(M, np.float64)
            average_power = 1
            nf_db = 5.5
            nf = 10.0 ** (nf_db / 10.0)
            amp_spam = 100
            amp_gain_db = 20
            amp_gain = 10.0 ** (amp_gain_db / 10.0)
            lambda_ = 1550
            h = 6.626068e-34
            f_hz = 2.99e8 / (lambda_ * 1e-9)
            for m, mod in enumerate(mods):
                SNR_min_calc = 10 ** ((mod.minimum_osnr + 2) / 10)
                for i, br in enumerate(table_rates):
                    v = (average_power * amp_spam) / (
                        SNR_min_calc * h * f_hz * amp_gain * nf * (br / mod.spectral_efficiency) * 1e9)
                    lmax_snr[m, i] = v / 1000
                lmax_xt[m] = 10 ** ((mod.inband_xt - worst_xt - 4) / 10)
        cum_class = class_reward = None
        if self.ENV_TYPE == 4:  # qos_constrained_ra.py:43-47, 262-265
            assert num_service_classes == len(classes_arrival_probabilities)
            self.num_service_classes = int(num_service_classes)
            self.classes_arrival_probabilities = list(classes_arrival_probabilities)
            self.classes_reward = list(classes_reward)
            cum_class = np.array(list(itertools.accumulate(classes_arrival_probabilities)), np.float64)
            class_reward = np.array(classes_reward, np.float64)
        path_mod = t.path_best_mod if modulations is None else t.path_modulation_for(mods)
        # ---- hand everything to the C ABI ----
        keep = dict(
            n_paths=np.ascontiguousarray(t.n_paths, np.int32), path_hops=np.ascontiguousarray(t.path_hops, np.int32),
            path_links=np.ascontiguousarray(t.path_links, np.int32),
            path_length=np.ascontiguousarray(t.path_length, np.float64),
            path_mod=np.ascontiguousarray(path_mod, np.int32),
            edge_iter_order=np.ascontiguousarray(t.edge_iter_order, np.int32),
            cum_src=cum_src, cum_dst=np.ascontiguousarray(cum_dst), bit_rates=np.array(table_rates, np.int32),
            cum_br=cum_br, n_slots=np.ascontiguousarray(n_slots), lmax_snr=lmax_snr, lmax_xt=lmax_xt,
            cum_class=cum_class, class_reward=class_reward)
        self._keep = keep
        desc = _lib.TopologyDesc(t.n_nodes, t.n_links, t.k_paths, t.max_hops, M, _ptr(keep["n_paths"]),
                                 _ptr(keep["path_hops"]), _ptr(keep["path_links"]), _ptr(keep["path_length"]),
                                 _ptr(keep["path_mod"]), _ptr(keep["edge_iter_order"]))
        cfg = _lib.EnvConfig(C.sizeof(_lib.EnvConfig), self.ENV_TYPE, num_spectrum_resources, num_spatial_resources, episode_length,
                             int(self.allow_rejection), j, mode, lo, hi, len(table_rates), event_capacity,
                             int(self.action_histograms),
                             lambda_a, lambda_h, _ptr(cum_src), _ptr(keep["cum_dst"]), _ptr(keep["bit_rates"]),
                             _ptr(cum_br), _ptr(keep["n_slots"]), _ptr(lmax_snr), _ptr(lmax_xt),
                             int(num_service_classes) if self.ENV_TYPE == 4 else 0, 0, _ptr(cum_class), _ptr(class_reward))
        if getattr(self, "_derive_only", False):  # spec_flags(): the configuration without a device
            self._cfg, self._desc, self._h, self._topo_h = cfg, desc, None, None
            return
        self._topo_h = C.c_void_p()
        self._ck(self.lib.orl_topology_create(C.byref(desc), device_id, C.byref(self._topo_h)))
        self._cfg, self._desc = cfg, desc  # (kept: what orl_multi_create takes to build the same envs over several devices)
        self._h = C.c_void_p()
        int_seeds = [41 if s_ is None else int(s_) for s_ in self.seeds]
        if all(-2**63 < s_ < 2**63 for s_ in int_seeds):
            # device-side random.Random(seed): no 2.5 KB/env upload, no Python loop over envs
            sd = np.array(int_seeds, np.int64)
            self._ck(self.lib.orl_batch_create_seeded(C.byref(cfg), self._topo_h, self.num_envs, sd.ctypes.data,
                                                        C.byref(self._h)))
        else:  # seeds beyond 64 bits: let CPython expand them
            st = mt_states(self.seeds)
            self._ck(self.lib.orl_batch_create(C.byref(cfg), self._topo_h, self.num_envs, st.ctypes.data,
                                                 C.byref(self._h)))
        self.n_info = self.lib.orl_batch_info_dim(self._h)
        self.obs_dim = self.lib.orl_batch_obs_dim(self._h)
        self.specialised = self._attach_specialisation()
        n = self.num_envs
        # host-side I/O arrays in page-locked memory: every step() moves actions in and reward/done/info(/obs) out
        self._act = self._host_array((n, 4), np.int32)
        self._act_in = self._host_array((n, 4), np.int32)
        self._reward = self._host_array((n,), np.float64)
        self._done = self._host_array((n,), np.uint8)
        self._info = self._host_array((n, self.n_info), np.float64)
        self._obs = self._host_array((n, self.obs_dim), np.float64) if self.obs_dim else None

    # ---- the persistent kernel with this configuration's sizes as compile-time constants ------------------------------
    JIT_MIN_ENVS = 4096

    def _attach_specialisation(self):
        """Attach the specialisation library of this configuration (include/orl.h, orl_batch_load_spec): a cached one whenever
        it exists; built on first use (~15 s of hipcc, once per configuration and source state) for batches of at least
        JIT_MIN_ENVS envs — where 5-12 % of the device loop are worth it — or whenever ORL_JIT_SPEC=1; ORL_JIT_SPEC=0: never.
        Returns whether one is attached.  Default library only (the cross-implementation builds keep the generic kernels)."""
        mode = os.environ.get("ORL_JIT_SPEC", "")
        variant = os.environ.get("ORL_LIB_VARIANT", "default")
        if mode == "0" or variant not in ("default", "exp") or (_build._extra() and variant != "exp"):
            return False
        buf = C.create_string_buffer(1024)
        if self.lib.orl_batch_spec_flags(self._h, buf, len(buf)) <= 0:
            return False
        flags = buf.value.decode()
        if os.environ.get("ORL_SPEC_EXTRA"):  # A/B experiments on the specialised kernels only: extra compiler flags (part of the cache key)
            flags += " " + os.environ["ORL_SPEC_EXTRA"]
        path = _build.spec_path(flags)
        from_cache = True
        if os.environ.get("ORL_SPEC_LIB"):  # A/B: a specialisation library built elsewhere (e.g. from another commit's device code)
            path = os.environ["ORL_SPEC_LIB"]
            from_cache = False  # (the user's file: never deleted, whatever the library says about it)
        if not os.path.exists(path):
            if not (mode == "1" or self.num_envs >= self.JIT_MIN_ENVS):
                return False
            try:
                path = _build.build_spec(flags)
            except Exception as exc:  # no compiler on this machine: the generic kernel runs
                import warnings

                warnings.warn("optical_rl_gym_amd: specialisation not built (%s); the generic persistent kernel runs" % exc)
                return False
        try:
            self._ck(self.lib.orl_batch_load_spec(self._h, path.encode()))
        except _lib.OrlError as exc:  # a cached library that no longer loads (another ROCm, a damaged file): optional, never fatal
            import warnings

            warnings.warn("optical_rl_gym_amd: specialisation %s not attached (%s); the generic persistent kernel runs"
                          % (os.path.basename(path), exc))
            # a cache entry that cannot be opened or is not a specialisation library is dropped (it would fail again every time);
            # one that is merely for another configuration or layout stays, and so does anything the user named
            if from_cache and any(t in str(exc) for t in ("dlopen", "not a specialisation", "other sources")):
                try:
                    os.unlink(path)
                except OSError:
                    pass
            return False
        return True

    @classmethod
    def spec_flags(cls, batch=1 << 20, **kwargs):
        """The -D flags of this configuration's specialisation for a batch of `batch` envs (the kernel form depends on it),
        computed without a device (pre-building, __graft_entry__.build)."""
        self = cls.__new__(cls)
        self._derive_only = True
        self.__init__(num_envs=1, **kwargs)
        buf = C.create_string_buffer(1024)
        n = self.lib.orl_spec_flags_for_batch(C.byref(self._cfg), C.byref(self._desc), int(batch), buf, len(buf))
        return buf.value.decode() if n > 0 else None

    def _ck(self, rc):
        _lib.check(rc, self.lib)

    def host_array(self, shape, dtype):
        """A page-locked numpy array (copies to and from it run at the full PCIe rate), e.g. for `step(obs_out=...)`."""
        return self._host_array(shape, dtype)

    def _host_array(self, shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        self._ck(self.lib.orl_host_alloc(max(nbytes, 1), C.byref(ptr)))
        buf = (C.c_ubyte * max(nbytes, 1)).from_address(ptr.value)
        buf._block = _PinnedBlock(self.lib, ptr)  # freed when the last numpy view of it is gone, not at close()
        return np.frombuffer(buf, dtype=np.uint8, count=nbytes).view(dtype).reshape(shape)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.orl_batch_destroy(self._h)
            self._h = None
        if getattr(self, "_topo_h", None):
            self.lib.orl_topology_destroy(self._topo_h)
            self._topo_h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the gym surface, batched ---------------------------------------------------------------
    def reset(self, full=False, mask=None):
        """reset(only_episode_counters = not full) for the envs selected by `mask` (default: all)."""
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._ck(self.lib.orl_batch_reset(self._h, int(full), _ptr(m)))
        return self.observation() if self.obs_dim else None

    def set_paths(self, paths):
        """The path each env's agent chose (PathOnlyFirstFitAction's Discrete(k + reject) action) for policy "PATH_FF"."""
        p = np.ascontiguousarray(np.asarray(paths).reshape(self.num_envs), np.int32)
        self._ck(self.lib.orl_batch_set_paths(self._h, p.ctypes.data))

    def policy(self, policy, fetch=True, paths=None):
        """On-device heuristic; returns [num_envs, 4] int32 actions (or None with fetch=False: they stay on
        the GPU for the next step(None)).  policy "PATH_FF" (PathOnlyFirstFitAction, rmsa_env.py:840-874 /
        rwa_env.py:505-536) takes the agents' path choices in `paths` (or from a previous set_paths / the "paths"
        device array)."""
        pid = POLICIES[policy] if isinstance(policy, str) else int(policy)
        if paths is not None:
            self.set_paths(paths)
        self._ck(self.lib.orl_batch_policy(self._h, pid, self._act.ctypes.data if fetch else None))
        return self._act if fetch else None

    def step(self, actions, auto_reset=False, fetch=True, obs_out=None, fetch_info=True):
        """actions: [num_envs, n_action] ints, or None to use the device-resident result of policy(fetch=False).
        Returns (obs, reward, done, info) arrays; info is [num_envs, n_info] in `info_keys` order.  The arrays are this
        object's page-locked staging buffers, overwritten by the next step.  `obs_out`: a caller-owned [num_envs, obs_dim]
        float64 or float32 array (ideally from `host_array`) that receives the observation instead — float32 is cast on the
        device (half the PCIe bytes, no host pass).  `fetch_info=False`: info stays on the device (None is returned for it);
        `info_rows(indices)` reads the rows that are needed afterwards."""
        if fetch:
            # the same call in its two halves (orl_batch_step_async / _wait): the action rows are checked and widened in ONE pass
            # inside the library instead of a strided numpy copy here and a second pass there (0.33 -> 0.2 ms at 65 536 envs)
            self.step_async(actions, auto_reset=auto_reset, obs_out=obs_out, fetch_info=fetch_info)
            return self.step_wait()
        a = None
        if actions is not None:
            actions = np.asarray(actions)
            if actions.ndim == 1:
                actions = actions[:, None]
            a = self._act_in  # columns beyond the family's action width stay zero from allocation
            a[:, : actions.shape[1]] = actions
        self._ck(self.lib.orl_batch_step(self._h, _ptr(a), int(auto_reset), None, None, None, None))
        return None

    def policy_step(self, policy, auto_reset=False, fetch=True, paths=None):
        """policy(policy) and step() on its actions in one call (include/orl.h, orl_batch_policy_step): ONE launch where the
        8-lanes-per-env step kernel serves the batch — the slot scan is its first phase.  Returns (actions, obs, reward, done,
        info) like policy() / step(), or None with fetch=False (everything stays on the GPU; nothing is synchronised)."""
        pid = POLICIES[policy] if isinstance(policy, str) else int(policy)
        if paths is not None:
            self.set_paths(paths)
        if not fetch:
            self._ck(self.lib.orl_batch_policy_step(self._h, pid, int(auto_reset), None, None, None, None, None))
            return None
        self._ck(self.lib.orl_batch_policy_step(self._h, pid, int(auto_reset), self._act.ctypes.data, _ptr(self._obs), self._reward.ctypes.data,
                                                  self._done.ctypes.data, self._info.ctypes.data))
        return self._act, self._obs, self._reward, self._done, self._info

    def set_info_mode(self, rates_only):
        """rates_only=True: the 8-lanes-per-env step kernel writes the blocking rates of info only (include/orl.h,
        orl_batch_set_info_mode); the compactness entries and the two link means keep whatever was written before."""
        self._ck(self.lib.orl_batch_set_info_mode(self._h, 1 if rates_only else 0))

    def action_bounds(self):
        """Exclusive upper bound per action column (None: any integer) — the index ranges of the reference's actions_output
        arrays that orl_batch_step checks before it modifies anything (rmsa_env.py:126-137, 167; rwa_env.py:52-58, 103;
        rmcsa_env.py:145-153, 219; qos_constrained_ra.py:101; DeepRMSA decodes any integer, deeprmsa_env.py:48-58)."""
        K, S, rej = self.k_paths, self.num_spectrum_resources, 1 if self.allow_rejection else 0
        t = self.ENV_TYPE
        if t == 0:
            return (K + 1, S + 1)
        if t == 2:
            return (K + rej, S + rej)
        if t == 3:
            return (K + 1, len(self.modulation_formats) + 1, self.num_spatial_resources + 1, S + 1)
        if t == 4:
            return (K + rej,)
        return (None,)

    def validate_actions(self, actions):
        """Raises the IndexError step() would, without touching the batch (a MultiDeviceBatch checks every shard's slice before it
        queues a step on any of them)."""
        if actions is None:
            return
        a = np.asarray(actions)
        if a.ndim == 1:
            a = a[:, None]
        for c, hi in enumerate(self.action_bounds()[: a.shape[1]]):
            if hi is None:
                continue
            bad = np.flatnonzero((a[:, c] < 0) | (a[:, c] >= hi))
            if len(bad):
                raise IndexError("action %s of env %d is outside the action space" % (tuple(int(v) for v in a[bad[0]]), int(bad[0])))

    def step_sync_abi(self, actions, auto_reset=False):
        """The synchronous entry point `orl_batch_step` with every output, as a C caller would use it ([num_envs][4] int32 action
        rows; tests compare it with the two-halves path)."""
        a = self._act_in
        a[:] = 0
        actions = np.asarray(actions)
        if actions.ndim == 1:
            actions = actions[:, None]
        a[:, : actions.shape[1]] = actions
        self._ck(self.lib.orl_batch_step(self._h, a.ctypes.data, int(auto_reset), _ptr(self._obs), self._reward.ctypes.data,
                                           self._done.ctypes.data, self._info.ctypes.data))
        return self._obs, self._reward, self._done, self._info

    def step_async(self, actions, auto_reset=False, obs_out=None, fetch_info=True):
        """First half of step() (include/orl.h, orl_batch_step_async): checks the actions, queues the copies and the kernel on
        the batch's stream and returns at once.  `step_wait()` collects what `step()` would have returned."""
        a, width, esz = None, 0, 0
        if actions is not None:
            a = np.asarray(actions)
            if a.ndim == 1:
                a = a[:, None]
            if a.dtype not in (np.int32, np.int64) or not a.flags.c_contiguous:
                a = np.ascontiguousarray(a, np.int64)
            assert a.shape[0] == self.num_envs and 1 <= a.shape[1] <= 4
            width, esz = a.shape[1], a.dtype.itemsize
            self._pending_actions = a  # (kept alive until the call returns: the library copies it before it does)
        obs64 = obs32 = None
        if self.obs_dim:
            obs = self._obs if obs_out is None else obs_out
            assert obs.shape == (self.num_envs, self.obs_dim) and obs.flags.c_contiguous
            if obs.dtype == np.float32:
                obs32 = obs
            else:
                assert obs.dtype == np.float64
                obs64 = obs
        else:
            obs = None
        self._ck(self.lib.orl_batch_step_async(self._h, _ptr(a), width, esz, int(auto_reset), _ptr(obs64), _ptr(obs32), self._reward.ctypes.data,
                                                 self._done.ctypes.data, self._info.ctypes.data if fetch_info else None))
        # (a refused call — bad action, a step still pending — leaves what step_wait() returns as it was)
        self._pending_obs, self._pending_info = obs, (self._info if fetch_info else None)

    def step_wait(self):
        self._ck(self.lib.orl_batch_step_wait(self._h))
        return self._pending_obs, self._reward, self._done, self._pending_info

    def info_rows(self, indices):
        """Rows `indices` of the info array the last step left on the device: [len(indices), n_info] float64 (gathered on the
        device; a VecEnv needs the rows of the envs that just finished an episode, not 4 MB of info per step)."""
        idx = np.ascontiguousarray(indices, np.int64)
        out = np.empty((len(idx), self.n_info), np.float64)
        if len(idx):
            self._ck(self.lib.orl_batch_get_info_rows(self._h, idx.ctypes.data, len(idx), out.ctypes.data))
        return out

    def run(self, policy, n_steps, time_kernels=False):
        """n_steps x (policy; step with auto reset) without leaving the device; returns RunStats."""
        pid = POLICIES[policy] if isinstance(policy, str) else int(policy)
        st = _lib.RunStats()
        self._ck(self.lib.orl_batch_run(self._h, pid, int(n_steps), int(time_kernels), C.byref(st)))
        return st

    def sync(self):
        self._ck(self.lib.orl_batch_sync(self._h))

    # ---- evaluate_heuristic on the device (utils.py:103-141) ---------------------------------------------------
    def steps_per_episode(self):
        """RMSA / DeepRMSA / RMCSA episodes last episode_length - 1 steps (the soft reset counts the pending service again,
        rmsa_env.py:310-315), RWA episodes episode_length steps (rwa_env.py:135-136, 160)."""
        return self.episode_length if self.ENV_TYPE in (2, 4) else self.episode_length - 1

    def evaluate(self, policy, n_eval_episodes=10):
        """n_eval_episodes episodes of every env under the on-device heuristic `policy`, with the reference harness's
        accounting (reset -> loop until done -> sum of rewards): returns (episode_rewards [num_envs, n_eval_episodes],
        episode_lengths).  One device-resident run; the kernels log each finished episode."""
        n = int(n_eval_episodes)
        self._ck(self.lib.orl_batch_reset(self._h, 0, None))  # the harness's reset() before the first episode (soft)
        self._ck(self.lib.orl_batch_episode_log(self._h, n))
        L = self.steps_per_episode()
        counts = np.zeros(self.num_envs, np.int32)
        acc = np.zeros((self.num_envs, n), np.int32)
        try:
            # the harness stops at the last done without resetting: all steps but the last in one device-resident run (auto
            # reset between episodes = the harness's reset() at the start of the next one), the last one without auto reset
            if n * L > 1:
                self.run(policy, n * L - 1)
            self.policy(policy, fetch=False)
            self.step(None, auto_reset=False, fetch=False)
            self.check()
            self._ck(self.lib.orl_batch_get_episode_log(self._h, counts.ctypes.data, acc.ctypes.data))
            qos_rewards = None
            if self.ENV_TYPE == 4:  # class rewards: the kernels kept the float64 sums (qos_constrained_ra.py:131-136)
                qos_rewards = np.zeros((self.num_envs, n), np.float64)
                self._ck(self.lib.orl_batch_get_episode_rewards(self._h, qos_rewards.ctypes.data))
        finally:  # whatever happened above, the next run must not append to a log nobody reads
            self._ck(self.lib.orl_batch_episode_log(self._h, 0))
        assert (counts == n).all(), "every env finishes exactly n episodes in n * steps_per_episode steps"
        rewards = acc.astype(np.float64) if qos_rewards is None else qos_rewards
        if self.ENV_TYPE == 1:  # DeepRMSA: +1 accepted, -1 otherwise (deeprmsa_env.py:123-124)
            rewards = 2.0 * rewards - L
        return rewards, np.full((self.num_envs, n), L, np.int64)

    def check(self):
        """Synchronise and raise what the kernels flagged since the last report: IndexError for a device-resident action
        outside the action space (rmsa_env.py:167), OverflowError when an env ran out of pending-release slots."""
        self._ck(self.lib.orl_batch_check(self._h))

    def seed(self, seeds, mask=None):
        """seed(seed) of the selected envs (optical_network_env.py:205-210): env i continues with the stream of
        random.Random(seeds[i]); a scalar means seed + i.  Nothing else of the state changes."""
        if np.isscalar(seeds):
            seeds = [int(seeds) + i for i in range(self.num_envs)]
        sd = np.array([41 if x is None else int(x) for x in seeds], np.int64)
        assert sd.shape == (self.num_envs,)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._ck(self.lib.orl_batch_reseed(self._h, sd.ctypes.data, _ptr(m)))
        for i in range(self.num_envs):
            if m is None or m[i]:
                self.seeds[i] = int(sd[i])

    # ---- zero-copy device views (an agent on the same GPU: no PCIe in the loop) -------------------------
    _BUFFERS = {"actions": (0, "<i4", 4), "reward": (1, "<f8", 0), "done": (2, "|u1", 0), "info": (3, "<f8", -1),
                "obs": (4, "<f8", -2), "terminal_obs": (5, "<f8", -2), "paths": (6, "<i4", 0)}

    def device_array(self, name):
        """The batch's device-resident I/O array `name` as an object with `__cuda_array_interface__` (what
        `torch.as_tensor(x, device="cuda")` and CuPy consume without a copy).  Write actions into "actions", call
        `step(None, fetch=False)`, `sync()`, read "reward" / "done" / "info" / "obs" in place."""
        which, typestr, cols = self._BUFFERS[name]
        ptr, n = C.c_void_p(), C.c_int64()
        self._ck(self.lib.orl_batch_device_buffer(self._h, which, C.byref(ptr), C.byref(n)))
        cols = {-1: self.n_info, -2: self.obs_dim}.get(cols, cols)
        if n.value == 0 or not ptr.value:
            raise _lib.OrlError("this env family has no '%s' array" % name)
        shape = (self.num_envs, cols) if cols else (self.num_envs,)

        cai = {"shape": shape, "typestr": typestr, "data": (int(ptr.value), False), "version": 2, "strides": None}
        device_id, owner = self.device_id, self

        class _Raw:  # what torch.as_tensor consumes
            __cuda_array_interface__ = cai

        class _DeviceArray:
            """Device-resident array of the batch: `__cuda_array_interface__` (torch.as_tensor, CuPy) and DLPack
            (`torch.from_dlpack`, `cupy.from_dlpack`, jax): no copy either way; the batch owns the memory."""
            __cuda_array_interface__ = cai
            _owner = owner  # keeps the batch alive

            def __dlpack_device__(self):
                return (10, device_id)  # kDLROCM

            def __dlpack__(self, stream=None, **kw):
                import torch

                t = torch.as_tensor(_Raw(), device="cuda:%d" % device_id)
                t._orl_owner = owner
                return t.__dlpack__() if stream is None else t.__dlpack__(stream=stream)

        return _DeviceArray()

    def device_tensor(self, name):
        """`device_array(name)` wrapped as a torch tensor on this batch's GPU (no copy)."""
        import torch

        return torch.as_tensor(self.device_array(name), device="cuda:%d" % self.device_id)

    def stream_ptr(self):
        """The HIP stream (an integer hipStream_t) this batch queues its launches on."""
        p = C.c_void_p()
        self._ck(self.lib.orl_batch_stream(self._h, C.byref(p)))
        return int(p.value or 0)

    def torch_stream(self):
        """The batch's stream as a `torch.cuda.ExternalStream`: inside `with torch.cuda.stream(env.torch_stream()):` an agent's
        kernels and `step(None, fetch=False)` are ordered by the stream — no synchronisation between the policy network and
        the step kernel, the host only queues work."""
        import torch

        return torch.cuda.ExternalStream(self.stream_ptr(), device="cuda:%d" % self.device_id)

    def observation(self):
        if not self.obs_dim:
            return None
        self._ck(self.lib.orl_batch_observation(self._h, self._obs.ctypes.data))
        return self._obs

    # ---- state read-back -------------------------------------------------------------------------
    def services(self):
        out = np.zeros((self.num_envs, 6))
        self._ck(self.lib.orl_batch_get_services(self._h, out.ctypes.data))
        return out

    def counters(self):
        out = np.zeros((self.num_envs, 8), np.int64)
        self._ck(self.lib.orl_batch_get_counters(self._h, out.ctypes.data))
        return out

    def slots(self, env=0):
        out = np.zeros((self.num_spatial_resources, self.topology.n_links, self.num_spectrum_resources), np.uint8)
        self._ck(self.lib.orl_batch_get_slots(self._h, env, out.ctypes.data))
        return out

    def spectrum(self, env=0):
        """QoSConstrainedRA: topology.graph["available_spectrum"] (free units per link)."""
        out = np.zeros(self.topology.n_links, np.int32)
        self._ck(self.lib.orl_batch_get_spectrum(self._h, env, out.ctypes.data))
        return out

    def link_stats(self, env=0):
        out = np.zeros((4, self.topology.n_links))
        self._ck(self.lib.orl_batch_get_link_stats(self._h, env, out.ctypes.data))
        return out

    def net_stats(self, env=0):
        out = np.zeros(4)
        self._ck(self.lib.orl_batch_get_net_stats(self._h, env, out.ctypes.data))
        return out

    def slots_packed(self):
        """The slot maps of every env as the device keeps them: [num_envs, map_words] uint64, bit s of word s // 64 of a
        (core, link) row set = slot s free (`orl_batch_row_words` words per row)."""
        out = np.zeros((self.num_envs, self.lib.orl_batch_map_words(self._h)), np.uint64)
        self._ck(self.lib.orl_batch_get_slots_packed(self._h, out.ctypes.data))
        return out

    def link_stats_all(self):
        """[num_envs, 4, links]: utilization, external_fragmentation, compactness, last_update of every link of every env."""
        out = np.zeros((self.num_envs, 4, self.topology.n_links))
        self._ck(self.lib.orl_batch_get_link_stats_all(self._h, out.ctypes.data))
        return out

    def net_stats_all(self):
        """[num_envs, 4]: throughput, compactness, last_update, current_time."""
        out = np.zeros((self.num_envs, 4))
        self._ck(self.lib.orl_batch_get_net_stats_all(self._h, out.ctypes.data))
        return out

    def active(self):
        out = np.zeros(self.num_envs, np.int32)
        self._ck(self.lib.orl_batch_get_active(self._h, out.ctypes.data))
        return out

    def n_active(self, env=0):
        return int(self.active()[env])

    def flags(self):
        out = np.zeros(self.num_envs, np.int32)
        self._ck(self.lib.orl_batch_get_flags(self._h, out.ctypes.data))
        return out

    def action_histograms_of(self, env=0):
        """(actions_output, actions_taken) of env `env` as [k_paths+1, slots+1] int arrays (rmsa_env.py:126-137; RWA's
        arrays are the top-left [k+reject, slots+reject] corner, rwa_env.py:52-58).  Needs action_histograms=True."""
        K1, S1 = self.k_paths + 1, self.num_spectrum_resources + 1
        shape = (2, K1, S1)
        if self.ENV_TYPE == 3:  # RMCSA: [k+1, modulations+1, cores+1, slots+1] (rmcsa_env.py:145-180)
            shape = (2, K1, len(self.modulation_formats) + 1, self.num_spatial_resources + 1, S1)
        out = np.zeros(shape, np.int32)
        self._ck(self.lib.orl_batch_get_action_histograms(self._h, env, out.ctypes.data))
        return out[0].astype(np.int64), out[1].astype(np.int64)

    def pending(self, env=0):
        """The pending releases of env `env` (the reference's `_events` heap, unordered): (release_time[n],
        records[n, 6] = (src*N+dst, path index, initial slot, number of slots, core, bit rate))."""
        n = self.lib.orl_batch_get_pending(self._h, env, 0, None, None)
        if n < 0:
            self._ck(n)
        t = np.zeros(max(n, 1), np.float64)
        rec = np.zeros((max(n, 1), 6), np.int32)
        n2 = self.lib.orl_batch_get_pending(self._h, env, n, t.ctypes.data, rec.ctypes.data)
        if n2 < 0:
            self._ck(n2)
        return t[:n], rec[:n]

    def matrix_observation(self):
        """SimpleMatrixObservation of every env, built on the device: uint8 [num_envs, 2N + C*E*S]."""
        dim = self.lib.orl_batch_matrix_obs_dim(self._h)
        out = np.zeros((self.num_envs, dim), np.uint8)
        self._ck(self.lib.orl_batch_matrix_observation(self._h, out.ctypes.data))
        return out

    def get_state(self):
        """Opaque snapshot of the whole batch (bytes); restore with set_state()."""
        buf = np.zeros(self.lib.orl_batch_state_bytes(self._h), np.uint8)
        self._ck(self.lib.orl_batch_get_state(self._h, buf.ctypes.data))
        return buf

    def set_state(self, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        assert buf.size == self.lib.orl_batch_state_bytes(self._h)
        self._ck(self.lib.orl_batch_set_state(self._h, buf.ctypes.data))

    def totals(self):
        p, a = C.c_int64(), C.c_int64()
        self._ck(self.lib.orl_batch_totals(self._h, C.byref(p), C.byref(a)))
        return p.value, a.value


class BatchedRMSAEnv(BatchedOpticalEnv):
    """reference: RMSAEnv (rmsa_env.py:18-744); gym id "RMSA-v0"."""

    ENV_TYPE = 0

    def __init__(self, topology=None, num_envs=1, seeds=None, device_id=0, episode_length=1000, load=10,
                 mean_service_holding_time=10800.0, num_spectrum_resources=100, bit_rate_selection="continuous",
                 bit_rates=(10, 40, 100), bit_rate_probabilities=None, node_request_probabilities=None,
                 bit_rate_lower_bound=25.0, bit_rate_higher_bound=100.0, seed=None, allow_rejection=False,
                 reset=True, channel_width=12.5, event_capacity=0, action_histograms=False):
        if seeds is None and seed is not None:
            seeds = seed
        self._setup(topology, num_envs, seeds, device_id, episode_length=episode_length, load=load,
                    mean_service_holding_time=mean_service_holding_time,
                    num_spectrum_resources=num_spectrum_resources, allow_rejection=allow_rejection,
                    node_request_probabilities=node_request_probabilities, channel_width=channel_width,
                    bit_rate_selection=bit_rate_selection, bit_rates=bit_rates,
                    bit_rate_probabilities=bit_rate_probabilities, bit_rate_lower_bound=bit_rate_lower_bound,
                    bit_rate_higher_bound=bit_rate_higher_bound, event_capacity=event_capacity,
                    action_histograms=action_histograms)
        self.info_keys = list(RMSA_INFO_KEYS)
        if bit_rate_selection == "discrete":
            self.info_keys += ["bit_rate_blocking_%s" % b for b in bit_rates] + ["fairness"]


class BatchedDeepRMSAEnv(BatchedOpticalEnv):
    """reference: DeepRMSAEnv (deeprmsa_env.py:9-132); gym id "DeepRMSA-v0"."""

    ENV_TYPE = 1
    N_ACTION = 1

    def __init__(self, topology=None, num_envs=1, seeds=None, device_id=0, j=1, episode_length=1000,
                 mean_service_holding_time=25.0, mean_service_inter_arrival_time=0.1, num_spectrum_resources=100,
                 node_request_probabilities=None, seed=None, allow_rejection=False, event_capacity=0,
                 action_histograms=False):
        if seeds is None and seed is not None:
            seeds = seed
        self._setup(topology, num_envs, seeds, device_id, episode_length=episode_length,
                    load=mean_service_holding_time / mean_service_inter_arrival_time,  # deeprmsa_env.py:25
                    mean_service_holding_time=mean_service_holding_time,
                    num_spectrum_resources=num_spectrum_resources, allow_rejection=allow_rejection,
                    node_request_probabilities=node_request_probabilities, channel_width=12.5, j=j,
                    event_capacity=event_capacity, action_histograms=action_histograms)
        self.info_keys = list(RMSA_INFO_KEYS)


class BatchedRWAEnv(BatchedOpticalEnv):
    """reference: RWAEnv (rwa_env.py:15-400); gym id "RWA-v0"."""

    ENV_TYPE = 2

    def __init__(self, topology=None, num_envs=1, seeds=None, device_id=0, episode_length=1000, load=10,
                 mean_service_holding_time=10800.0, num_spectrum_resources=80, node_request_probabilities=None,
                 allow_rejection=True, seed=None, reset=True, channel_width=50.0, event_capacity=0,
                 action_histograms=False):
        if seeds is None and seed is not None:
            seeds = seed
        self._setup(topology, num_envs, seeds, device_id, episode_length=episode_length, load=load,
                    mean_service_holding_time=mean_service_holding_time,
                    num_spectrum_resources=num_spectrum_resources, allow_rejection=allow_rejection,
                    node_request_probabilities=node_request_probabilities, channel_width=channel_width,
                    event_capacity=event_capacity, action_histograms=action_histograms)
        rej = self.reject_action
        self.info_keys = (["service_blocking_rate", "episode_service_blocking_rate"]
                          + ["path_action_probability[%d]" % i for i in range(self.k_paths + rej)]
                          + ["wavelength_action_probability[%d]" % i for i in range(num_spectrum_resources + rej)])


class BatchedRMCSAEnv(BatchedOpticalEnv):
    """reference: RMCSAEnv (rmcsa_env.py:18-879); gym id "RMCSA-v0"."""

    ENV_TYPE = 3
    N_ACTION = 4

    def __init__(self, topology=None, num_envs=1, seeds=None, device_id=0, episode_length=1000, load=10,
                 mean_service_holding_time=10800.0, num_spectrum_resources=100, num_spatial_resources=7,
                 modulation_formats=None, worst_xt=None, node_request_probabilities=None,
                 bit_rate_selection="continuous", bit_rates=(10, 40, 100), bit_rate_probabilities=None,
                 bit_rate_lower_bound=25, bit_rate_higher_bound=100, seed=None, allow_rejection=False, reset=True,
                 channel_width=12.5, event_capacity=0, action_histograms=False):
        import copy

        if seeds is None and seed is not None:
            seeds = seed
        topo = Topology.load(topology) if isinstance(topology, str) else topology
        mods = copy.deepcopy(list(topo.modulations if modulation_formats is None else modulation_formats))
        if worst_xt is None:  # rmcsa_env.py:63-67, 119-122
            worst_xt = {7: -84.7, 12: -61.9, 19: -54.8}.get(num_spatial_resources)
        for m in mods:  # rmcsa_env.py:127-129: +4 dB margin on both limits
            m.inband_xt += 4
        worst_xt += 4
        self._setup(topo, num_envs, seeds, device_id, episode_length=episode_length, load=load,
                    mean_service_holding_time=mean_service_holding_time,
                    num_spectrum_resources=num_spectrum_resources, allow_rejection=allow_rejection,
                    node_request_probabilities=node_request_probabilities, channel_width=channel_width,
                    bit_rate_selection=bit_rate_selection, bit_rates=bit_rates,
                    bit_rate_probabilities=bit_rate_probabilities, bit_rate_lower_bound=bit_rate_lower_bound,
                    bit_rate_higher_bound=bit_rate_higher_bound, num_spatial_resources=num_spatial_resources,
                    modulations=mods, worst_xt=worst_xt, event_capacity=event_capacity, action_histograms=action_histograms)
        self.info_keys = RMSA_INFO_KEYS[:4]


class BatchedQoSConstrainedRA(BatchedOpticalEnv):
    """reference: QoSConstrainedRA (qos_constrained_ra.py:13-398); gym id "QoSConstrainedRA-v0".  Upstream its constructor
    raises (an unexpected `k_paths` keyword for the base class, :32-41, and a `service_class` field utils.Service lacks);
    the semantics here are the class's as written, pinned by fixtures captured from the reference with those two things
    repaired at import time (oracle/gen_golden_qos.py).  Actions: the path index; the service class of the pending service
    is column 4 of services()."""

    ENV_TYPE = 4
    N_ACTION = 1

    def __init__(self, topology=None, num_envs=1, seeds=None, device_id=0, episode_length=1000, load=10,
                 mean_service_holding_time=10800.0, num_spectrum_resources=80, num_service_classes=1,
                 classes_arrival_probabilities=(1.0,), classes_reward=(1.0,), node_request_probabilities=None,
                 allow_rejection=True, k_paths=5, seed=None, reset=True, event_capacity=0):
        if seeds is None and seed is not None:
            seeds = seed
        self._setup(topology, num_envs, seeds, device_id, episode_length=episode_length, load=load,
                    mean_service_holding_time=mean_service_holding_time,
                    num_spectrum_resources=num_spectrum_resources, allow_rejection=allow_rejection,
                    node_request_probabilities=node_request_probabilities, channel_width=12.5,
                    event_capacity=event_capacity, num_service_classes=num_service_classes,
                    classes_arrival_probabilities=classes_arrival_probabilities, classes_reward=classes_reward)
        self.info_keys = ["service_blocking_rate", "episode_service_blocking_rate"]


ENV_CLASSES = {"RMSA": BatchedRMSAEnv, "DeepRMSA": BatchedDeepRMSAEnv, "RWA": BatchedRWAEnv, "RMCSA": BatchedRMCSAEnv,
               "RMSA-v0": BatchedRMSAEnv, "DeepRMSA-v0": BatchedDeepRMSAEnv, "RWA-v0": BatchedRWAEnv,
               "RMCSA-v0": BatchedRMCSAEnv, "QoSConstrainedRA": BatchedQoSConstrainedRA,
               "QoSConstrainedRA-v0": BatchedQoSConstrainedRA}


def make(env_id, device_ids=None, **kwargs):
    """Batch constructor by registry id (optical_rl_gym/__init__.py:3-26).  `device_ids=[0, 1, ...]` spreads the envs over
    several GPUs of the node behind one object (sharding.MultiDeviceBatch); default: one batch on `device_id`."""
    if device_ids is not None and len(device_ids) > 1:
        from .sharding import MultiDeviceBatch

        kwargs.pop("device_id", None)
        return MultiDeviceBatch(env_id, kwargs.pop("num_envs"), seeds=kwargs.pop("seeds", None), device_ids=device_ids, **kwargs)
    if device_ids is not None and len(device_ids) == 1:
        kwargs["device_id"] = int(device_ids[0])
    return ENV_CLASSES[env_id](**kwargs)
