"""Single-env, gym.Env-shaped front ends with the reference's public surface.

`RMSAEnv(topology=..., seed=..., **kwargs)` etc. behave like the reference classes for the scripts in the
reference's tests/ directory: `reset()`, `step(action) -> (obs, reward, done, info dict)`, `seed()`,
`current_service`, counters (`services_processed`, `episode_bit_rate_requested`, ...), `actions_output` /
`actions_taken`, `topology.graph[...]` state views, `k_shortest_paths`, spaces, the query methods user heuristics
call (`is_path_free`, `get_available_slots`, `get_available_blocks`, `rle`, `get_path_capacity`), the module-level
heuristics and `evaluate_heuristic`.  The work is done by a 1-env batch of the HIP library (or by any object with the
same methods — the CPU tests plug the oracle in to exercise this file without a GPU).

reference: rmsa_env.py, deeprmsa_env.py, rwa_env.py, rmcsa_env.py, utils.py:37-59 (Service), :103-141
(evaluate_heuristic), wrappers rmsa_env.py:806-874, rwa_env.py:505-536, rmcsa_env.py:914-947.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import spaces
from .envs import (COUNTER_NAMES, RMSA_INFO_KEYS, BatchedDeepRMSAEnv, BatchedRMCSAEnv, BatchedRMSAEnv,
                   BatchedRWAEnv)
from .topology import Path, Topology


@dataclass(repr=False)
class Service:  # reference: utils.py:37-59
    service_id: int
    source: str
    source_id: int
    destination: Optional[str] = field(default=None)
    destination_id: Optional[int] = field(default=None)
    arrival_time: Optional[float] = field(default=None)
    holding_time: Optional[float] = field(default=None)
    bit_rate: Optional[float] = field(default=None)
    path: Optional[Path] = field(default=None)
    number_slots: Optional[int] = field(default=None)
    core: Optional[int] = field(default=None)
    accepted: bool = field(default=False)

    def __str__(self):
        msg = "{" + ("" if self.bit_rate is None else f"br: {self.bit_rate}, ")
        return f"Serv. {self.service_id} ({self.source} -> {self.destination})" + msg


class _KSP:
    """`env.k_shortest_paths[src_name, dst_name]` -> list of Path (topology.graph["ksp"])."""

    def __init__(self, topo):
        self.topo = topo
        self.index = {n: i for i, n in enumerate(topo.node_names)}

    def __getitem__(self, key):
        s, d = key
        return self.topo.ksp(self.index[s] if isinstance(s, str) else int(s), self.index[d] if isinstance(d, str) else int(d))


class _GraphView:
    """`env.topology.graph[key]`: live views of the device state under the reference's key names."""

    def __init__(self, env):
        self.env = env

    def __getitem__(self, key):
        e = self.env
        if key in ("available_slots", "available_wavelengths"):
            return e._slots()
        if key == "available_spectrum":  # QoSConstrainedRA (optical_network_env.py:189-193)
            return e.batch.spectrum(0).astype(int)
        if key == "throughput":
            return float(e.batch.net_stats(0)[0])
        if key == "compactness":
            return float(e.batch.net_stats(0)[1])
        if key == "last_update":
            return float(e.batch.net_stats(0)[2])
        if key == "num_spectrum_resources":
            return e.num_spectrum_resources
        if key == "k_paths":
            return e.k_paths
        if key == "ksp":
            return e.k_shortest_paths
        if key == "modulations":
            return e.topo.modulations
        if key == "node_indices":
            return e.topo.node_names
        if key == "name":
            return e.topo.name
        raise KeyError(key)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False


class _TopologyView:
    def __init__(self, env):
        self._env = env
        self.graph = _GraphView(env)

    def number_of_nodes(self):
        return self._env.topo.n_nodes

    def number_of_edges(self):
        return self._env.topo.n_links

    def nodes(self):
        return list(self._env.topo.node_names)

    def edges(self):
        t = self._env.topo
        return [(t.node_names[t.link_nodes[i][0]], t.node_names[t.link_nodes[i][1]]) for i in t.edge_iter_order]

    def link_attr(self, index):
        """utilization / external_fragmentation / compactness / last_update of link `index`."""
        ls = self._env.batch.link_stats(0)
        return dict(utilization=ls[0, index], external_fragmentation=ls[1, index], compactness=ls[2, index],
                    last_update=ls[3, index])


class _SingleEnv:
    """Shared implementation of the four gym-shaped classes."""

    BATCH_CLS = None
    POLICY_SAP = "SAP_FF"
    metadata = {"metrics": ["service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate",
                            "episode_bit_rate_blocking_rate"]}

    def __init__(self, topology=None, seed=None, _backend=None, **kwargs):
        self.rand_seed = 41 if seed is None else seed
        if _backend is None:
            # (RMCSA's 4-D arrays are 860 KB per env at 7 x 320: affordable for the 1-env front end, opt-in for batches)
            extra = dict(action_histograms=True) if self.BATCH_CLS.ENV_TYPE in (0, 1, 2, 3) else {}
            _backend = self.BATCH_CLS(topology=topology, num_envs=1, seeds=[seed], **extra, **kwargs)
            topo = _backend.topology
        else:
            topo = Topology.load(topology) if isinstance(topology, str) else topology
        self.batch = _backend
        self.topo = topo
        self.topology = _TopologyView(self)
        self.topology_name = topo.name
        self.k_paths = topo.k_paths
        self.k_shortest_paths = _KSP(topo)
        self._link_of = {}
        for i, (a, b) in enumerate(topo.link_nodes):
            self._link_of[(topo.node_names[a], topo.node_names[b])] = i
            self._link_of[(topo.node_names[b], topo.node_names[a])] = i
        self.num_spectrum_resources = kwargs.get("num_spectrum_resources", self.DEFAULT_SLOTS)
        self.episode_length = kwargs.get("episode_length", 1000)
        self.channel_width = kwargs.get("channel_width", self.DEFAULT_CHANNEL_WIDTH)
        self.allow_rejection = kwargs.get("allow_rejection", self.DEFAULT_REJECTION)
        self.reject_action = 1 if self.allow_rejection else 0
        self._accepted = False
        self._episode_base = None  # action histograms at the last reset (episode_actions_* = total - base)
        self._make_spaces(kwargs)
        self.action_space.seed(self.rand_seed)
        self.observation_space.seed(self.rand_seed)

    DEFAULT_SLOTS = 100
    DEFAULT_REJECTION = False
    DEFAULT_CHANNEL_WIDTH = 12.5

    def _make_spaces(self, kwargs):
        self.action_space = spaces.MultiDiscrete((self.k_paths + self.reject_action,
                                                  self.num_spectrum_resources + self.reject_action))
        self.observation_space = spaces.Dict({"topology": spaces.Discrete(10), "current_service": spaces.Discrete(10)})

    @property
    def unwrapped(self):
        return self

    # ---- counters (optical_network_env.py:29-34, rmsa_env.py:73-76) ----
    def __getattr__(self, name):
        if name in COUNTER_NAMES:
            return int(self.batch.counters()[0, COUNTER_NAMES.index(name)])
        raise AttributeError(name)

    @property
    def current_time(self):
        return float(self.batch.net_stats(0)[3])

    @property
    def current_service(self):
        at, ht, src, dst, br, sid = self.batch.services()[0]
        names = self.topo.node_names
        return Service(int(sid), names[int(src)], int(src), names[int(dst)], int(dst), float(at), float(ht),
                       int(br) if self.BATCH_CLS is not BatchedRWAEnv else None, accepted=self._accepted)

    service = current_service  # RMCSAEnv's wrappers spell it `env.service` (rmcsa_env.py:965)

    # ---- 2-D action histograms (rmsa_env.py:126-137, rwa_env.py:52-58) ----
    def _histograms(self):
        out, taken = self.batch.action_histograms_of(0)
        return out, taken

    @property
    def actions_output(self):
        return self._histograms()[0]

    @property
    def actions_taken(self):
        return self._histograms()[1]

    def _episode_hist(self, which):
        k, s = self.k_paths + self.reject_action, self.num_spectrum_resources + self.reject_action
        return np.zeros((k, s), dtype=int)  # RMSAEnv re-zeroes them at reset and never increments them (rmsa_env.py:289-302)

    @property
    def episode_actions_output(self):
        return self._episode_hist(0)

    @property
    def episode_actions_taken(self):
        return self._episode_hist(1)

    # ---- gym surface ----
    def observation(self):
        return {"topology": self.topology, "service": self.current_service}

    def reset(self, only_episode_counters=True, only_counters=None):
        if only_counters is not None:  # RWAEnv spells the argument differently (rwa_env.py:164)
            only_episode_counters = only_counters
        self.batch.reset(full=not only_episode_counters)
        self._episode_base = None
        return self.observation()

    def _encode(self, action):
        return np.atleast_2d(np.asarray(action, dtype=np.int64))

    def step(self, action):
        """IndexError for an action outside actions_output's shape, before anything changes (rmsa_env.py:167)."""
        _, reward, done, info = self.batch.step(self._encode(action))
        self._accepted = bool(reward[0] > 0)
        return self.observation(), self._reward_value(reward[0]), bool(done[0]), self._info_dict(info[0])

    def _reward_value(self, r):
        return int(r)

    def _info_dict(self, row):
        keys = getattr(self.batch, "info_keys", None) or RMSA_INFO_KEYS
        return {k: float(v) for k, v in zip(keys, row)}

    def render(self, mode="human"):
        return

    def seed(self, seed=None):
        """optical_network_env.py:205-210: the env continues with random.Random(seed or 41).  (As in the reference, bit
        rates keep coming from the Random object the constructor bound: rmsa_env.py:85-87.)"""
        self.rand_seed = 41 if seed is None else seed
        self.batch.seed([self.rand_seed])
        return [self.rand_seed]

    def close(self):
        if hasattr(self.batch, "close"):
            self.batch.close()

    # ---- queries user heuristics call; each reads the env's slot map once ----
    def _slots(self):
        sl = self.batch.slots(0).astype(int)
        return sl if sl.shape[0] > 1 else sl[0]

    def _links(self, path):
        return [self._link_of[(path.node_list[i], path.node_list[i + 1])] for i in range(len(path.node_list) - 1)]

    def get_number_slots(self, path, modulation=None):
        """rmsa_env.py:610-621 / rmcsa_env.py:753-765 (guard band included)."""
        import math

        mod = modulation if modulation is not None else path.best_modulation
        return math.ceil(self.current_service.bit_rate / (mod.spectral_efficiency * self.channel_width)) + 1

    def is_path_free(self, path, initial_slot, number_slots):
        """rmsa_env.py:623-636"""
        if initial_slot + number_slots > self.num_spectrum_resources:
            return False
        avail = self._slots()
        return not np.any(avail[self._links(path), initial_slot:initial_slot + number_slots] == 0)

    def get_available_slots(self, path):
        """rmsa_env.py:638-649 (rows by link index; the reference's `"id"` lookup only works where id == index)."""
        return np.prod(self._slots()[self._links(path), :], axis=0)

    @staticmethod
    def rle(inarray):
        """Run-length encoding (rmsa_env.py:651-665): (start positions, run values, run lengths)."""
        ia = np.asarray(inarray)
        n = len(ia)
        if n == 0:
            return None, None, None
        change = np.flatnonzero(ia[1:] != ia[:-1])
        ends = np.append(change, n - 1)
        lengths = np.diff(np.append(-1, ends))
        starts = np.cumsum(np.append(0, lengths))[:-1]
        return starts, ia[ends], lengths

    def get_available_blocks(self, path_index):
        """rmsa_env.py:667-697: the first j free blocks of path `path_index` that fit the pending service."""
        svc = self.current_service
        path = self.k_shortest_paths[svc.source, svc.destination][path_index]
        available = self.get_available_slots(path)
        slots = self.get_number_slots(path)
        starts, values, lengths = self.rle(available)
        ok = np.flatnonzero((values == 1) & (lengths >= slots))[: self.j]
        return starts[ok], lengths[ok]

    def policy_action(self, policy):
        """Action of the on-device heuristic `policy` for the pending service, in the reference's tuple form."""
        a = self.batch.policy(policy)[0]
        return self._decode(a)

    def _decode(self, a):
        return (int(a[0]), int(a[1]))


class RMSAEnv(_SingleEnv):
    BATCH_CLS = BatchedRMSAEnv


class DeepRMSAEnv(_SingleEnv):
    BATCH_CLS = BatchedDeepRMSAEnv
    POLICY_SAP = "SAP"

    def __init__(self, topology=None, seed=None, _backend=None, **kwargs):
        self.j = kwargs.get("j", 1)
        super().__init__(topology, seed, _backend, **kwargs)

    def _make_spaces(self, kwargs):
        n = self.topo.n_nodes
        shape = 1 + 2 * n + (2 * self.j + 3) * self.k_paths
        self.observation_space = spaces.Box(low=-2**30, high=2**30, dtype=np.float64, shape=(shape,))
        self.action_space = spaces.Discrete(self.k_paths * self.j + self.reject_action)

    def observation(self):
        return np.array(self.batch.observation()[0])

    def _encode(self, action):
        return np.array([[int(action)]], dtype=np.int64)

    def _decode(self, a):
        return int(a[0])

    def _get_route_block_id(self, action):
        """deeprmsa_env.py:126-129"""
        return action // self.j, action % self.j


class RWAEnv(_SingleEnv):
    BATCH_CLS = BatchedRWAEnv
    DEFAULT_SLOTS = 80
    DEFAULT_REJECTION = True
    DEFAULT_CHANNEL_WIDTH = 50.0
    metadata = {"metrics": ["service_blocking_rate", "episode_service_blocking_rate"]}

    def _info_dict(self, row):
        npa = self.k_paths + self.reject_action
        return {"service_blocking_rate": float(row[0]), "episode_service_blocking_rate": float(row[1]),
                "path_action_probability": np.array(row[2:2 + npa]),
                "wavelength_action_probability": np.array(row[2 + npa:])}

    def is_path_free(self, path, wavelength):
        """rwa_env.py:385-400"""
        if wavelength > self.num_spectrum_resources:
            return False
        avail = self._slots()
        return not np.any(avail[self._links(path), wavelength] == 0)

    def get_path_capacity(self, path):
        """rwa_env.py:403-422: wavelengths free on every link of the path."""
        return int(np.sum(np.prod(self._slots()[self._links(path), :], axis=0)))

    @property
    def actions_output(self):  # [k + reject][S + reject] (rwa_env.py:52-58)
        k, s = self.k_paths + self.reject_action, self.num_spectrum_resources + self.reject_action
        return self._histograms()[0][:k, :s]

    def reset(self, only_counters=True, only_episode_counters=None):
        out = super().reset(only_episode_counters=only_counters if only_episode_counters is None else only_episode_counters)
        self._episode_base = tuple(h.copy() for h in self._histograms())  # RWAEnv does count per episode (rwa_env.py:104, 126)
        return out

    def _episode_hist(self, which):
        cur = self._histograms()[which]
        base = np.zeros_like(cur) if self._episode_base is None else self._episode_base[which]
        d = cur - base
        if which == 0:
            return d[: self.k_paths + self.reject_action, : self.num_spectrum_resources + self.reject_action]
        return d


class RMCSAEnv(_SingleEnv):
    BATCH_CLS = BatchedRMCSAEnv
    POLICY_SAP = "SAP_BM_FC_FF"

    def _make_spaces(self, kwargs):
        self.num_spatial_resources = kwargs.get("num_spatial_resources", 7)
        n_mod = len(self.topo.modulations)
        self.action_space = spaces.MultiDiscrete((self.k_paths + self.reject_action, n_mod,
                                                  self.num_spatial_resources + self.reject_action,
                                                  self.num_spectrum_resources + self.reject_action))
        self.observation_space = spaces.Dict({"topology": spaces.Discrete(10), "current_service": spaces.Discrete(10)})

    def _decode(self, a):
        return (int(a[0]), int(a[1]), int(a[2]), int(a[3]))

    def is_path_free(self, core, path, initial_slot, number_slots):
        """rmcsa_env.py:767-794"""
        if initial_slot + number_slots > self.num_spectrum_resources:
            return False
        avail = self._slots()
        return not np.any(avail[core][self._links(path), initial_slot:initial_slot + number_slots] == 0)

    def _episode_hist(self, which):  # re-zeroed at every reset and never incremented (rmcsa_env.py:154-180, 391-407)
        return np.zeros((self.k_paths + 1, len(self.topo.modulations) + 1, self.num_spatial_resources + 1,
                         self.num_spectrum_resources + 1), dtype=int)


# ---- module-level heuristics with the reference's names ---------------------------------------------------
def _device_heuristic(policy):
    def wrap(fn):
        fn.device_policy = policy  # evaluate_heuristic runs these without leaving the GPU
        return fn
    return wrap


@_device_heuristic("SP_FF")
def shortest_path_first_fit(env):
    """rmsa_env.py:747-764 / deeprmsa_env.py:135-143 / rwa_env.py:425-435"""
    return env.unwrapped.policy_action("SP_FF")


@_device_heuristic("SAP_FF")
def shortest_available_path_first_fit(env):
    """rmsa_env.py:767-779 / deeprmsa_env.py:146-155 / rwa_env.py:438-457"""
    return env.unwrapped.policy_action("SAP_FF")


@_device_heuristic("LLP_FF")
def least_loaded_path_first_fit(env):
    """rmsa_env.py:782-803 / rwa_env.py:482-502"""
    return env.unwrapped.policy_action("LLP_FF")


@_device_heuristic("SAP_LF")
def shortest_available_path_last_fit(env):
    """rwa_env.py:460-479"""
    return env.unwrapped.policy_action("SAP_LF")


@_device_heuristic("SAP_BM_FC_FF")
def shortest_available_path_best_modulation_first_core_first_fit(env):
    """rmcsa_env.py:882-911"""
    return env.unwrapped.policy_action("SAP_BM_FC_FF")


def random_policy(env):
    """utils.py:99-100 (the stream is this package's numpy generator, not gym 0.21's)"""
    return env.action_space.sample()


def start_environment(env, steps):
    """utils.py:62-70.  As written upstream the episode loop never runs: `done` starts True, every one of the `steps`
    iterations resets the env (soft reset) and the inner `while not done` is skipped, so the function returns the env after
    `steps` soft resets.  Kept with exactly that observable behaviour; an episode that did start is played with random
    actions until done, as upstream intends."""
    done = True
    for _ in range(int(steps)):
        if done:
            env.reset()
        while not done:
            _, _, done, _ = env.step(env.action_space.sample())
    return env


def evaluate_heuristic(env, heuristic, n_eval_episodes=10, render=False, callback=None, reward_threshold=None,
                       return_episode_rewards=False):
    """Episode returns of `heuristic` on `env` with the accounting of utils.py:103-141 (soft reset, play until done, sum
    the rewards; mean / std over the episodes, or the per-episode lists).

    `env` is a 1-env front end (RMSAEnv, ...) or a batch (BatchedRMSAEnv, ...).  The heuristics of this module run
    entirely on the device: one run of n_eval_episodes x steps-per-episode steps, the kernels log every finished episode
    (`BatchedOpticalEnv.evaluate`); for a batch the result has one row per env.  Any other callable (and render / callback)
    goes through the host loop, one step() per decision."""
    policy = getattr(heuristic, "device_policy", heuristic if isinstance(heuristic, str) else None)
    batch = env if hasattr(env, "evaluate") and not hasattr(env, "unwrapped") else None
    single = getattr(env, "unwrapped", None) if batch is None else None
    on_device = policy is not None and not render and callback is None and \
        (batch is not None or (single is env and hasattr(getattr(single, "batch", None), "evaluate")))
    if on_device:
        b = batch if batch is not None else single.batch
        if policy in ("SAP_FF", "SP_FF") and b.ENV_TYPE == 1:
            policy = "SAP" if policy == "SAP_FF" else "SP"
        rewards, lengths = b.evaluate(policy, n_eval_episodes)
        if batch is None:
            rewards, lengths = rewards[0], lengths[0]
            single._accepted = False
    else:
        if batch is not None:
            raise TypeError("a batch is evaluated with one of the on-device heuristics")
        returns, steps = [], []
        while len(returns) < n_eval_episodes:
            env.reset()
            total, count, finished = 0.0, 0, False
            while not finished:
                _, r, finished, _ = env.step(heuristic(env))
                total += r
                count += 1
                if callback is not None:
                    callback(locals(), globals())
                if render:
                    env.render()
            returns.append(total)
            steps.append(count)
        rewards, lengths = np.asarray(returns, np.float64), np.asarray(steps, np.int64)
    mean_reward, std_reward = np.mean(rewards, axis=-1), np.std(rewards, axis=-1)
    if reward_threshold is not None:
        assert np.all(mean_reward > reward_threshold), "Mean reward below threshold: {} < {:.2f}".format(mean_reward, reward_threshold)
    if return_episode_rewards:
        return rewards.tolist(), lengths.tolist()
    return mean_reward, std_reward


# ---- wrappers ---------------------------------------------------------------------------------------------
class _Wrapper:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)


class SimpleMatrixObservation(_Wrapper):
    """rmsa_env.py:806-837, rmcsa_env.py:914-947: [one-hot(min(src,dst)), one-hot(max(src,dst)), slot map], built by the
    device (k_matrix_obs) and returned with the reference's dtype (float64 values in a uint8-declared Box)."""

    def __init__(self, env):
        super().__init__(env)
        u = env.unwrapped
        cores = getattr(u, "num_spatial_resources", 1)
        shape = u.topo.n_nodes * 2 + u.topo.n_links * u.num_spectrum_resources * cores
        self.observation_space = spaces.Box(low=0, high=1, dtype=np.uint8, shape=(shape,))
        self.action_space = env.action_space

    def observation(self, observation=None):
        u = self.env.unwrapped
        return np.asarray(u.batch.matrix_observation()[0], np.float64).reshape(self.observation_space.shape)

    def reset(self, **kw):
        self.env.reset(**kw)
        return self.observation()

    def step(self, action):
        _, r, d, i = self.env.step(action)
        return self.observation(), r, d, i


class PathOnlyFirstFitAction(_Wrapper):
    """rmsa_env.py:840-874 / rwa_env.py:505-536: the agent picks the path, the device finds the first fitting slot on it
    (policy "PATH_FF": the reference's loop incl. its `range(0, S - n)` bound for RMSA, all wavelengths for RWA)."""

    def __init__(self, env):
        super().__init__(env)
        u = env.unwrapped
        self.action_space = spaces.Discrete(u.k_paths + u.reject_action)
        self.observation_space = env.observation_space

    def action(self, action):
        u = self.env.unwrapped
        a = u.batch.policy("PATH_FF", paths=[int(action)])[0]
        return (int(a[0]), int(a[1]))

    def step(self, action):
        return self.env.step(self.action(action))
