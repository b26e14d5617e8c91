"""Single-env, gym.Env-shaped front ends with the reference's public surface.

`RMSAEnv(topology=..., seed=..., **kwargs)` etc. behave like the reference classes for the scripts in the
reference's tests/ directory: `reset()`, `step(action) -> (obs, reward, done, info dict)`, `current_service`,
counters (`services_processed`, `episode_bit_rate_requested`, ...), `topology.graph[...]` state views,
`k_shortest_paths`, spaces, the module-level heuristics and `evaluate_heuristic`.  The work is done by a
1-env batch of the HIP library (or by any object with the same methods — the CPU tests plug the oracle in
to exercise this file without a GPU).

reference: rmsa_env.py, deeprmsa_env.py, rwa_env.py, rmcsa_env.py, utils.py:37-59 (Service), :103-141
(evaluate_heuristic), wrappers rmsa_env.py:806-874, rwa_env.py:505-536.
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
            sl = e.batch.slots(0).astype(int)
            return sl if e.batch.num_spatial_resources > 1 else sl[0]
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
    metadata = {"metrics": ["service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate",
                            "episode_bit_rate_blocking_rate"]}

    def __init__(self, topology=None, seed=None, _backend=None, **kwargs):
        self.rand_seed = 41 if seed is None else seed
        if _backend is None:
            _backend = self.BATCH_CLS(topology=topology, num_envs=1, seeds=[seed], **kwargs)
            topo = _backend.topology
        else:
            topo = Topology.load(topology) if isinstance(topology, str) else topology
        self.batch = _backend
        self.topo = topo
        self.topology = _TopologyView(self)
        self.topology_name = topo.name
        self.k_paths = topo.k_paths
        self.k_shortest_paths = _KSP(topo)
        self.num_spectrum_resources = kwargs.get("num_spectrum_resources", self.DEFAULT_SLOTS)
        self.episode_length = kwargs.get("episode_length", 1000)
        self.channel_width = kwargs.get("channel_width", self.DEFAULT_CHANNEL_WIDTH)
        self.allow_rejection = kwargs.get("allow_rejection", self.DEFAULT_REJECTION)
        self.reject_action = 1 if self.allow_rejection else 0
        self._accepted = False
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

    # ---- gym surface ----
    def observation(self):
        return {"topology": self.topology, "service": self.current_service}

    def reset(self, only_episode_counters=True, only_counters=None):
        if only_counters is not None:  # RWAEnv spells the argument differently (rwa_env.py:164)
            only_episode_counters = only_counters
        self.batch.reset(full=not only_episode_counters)
        return self.observation()

    def _encode(self, action):
        return np.atleast_2d(np.asarray(action, dtype=np.int64))

    def step(self, action):
        _, reward, done, info = self.batch.step(self._encode(action))
        self._accepted = bool(reward[0] > 0)
        return self.observation(), self._reward_value(reward[0]), bool(done[0]), self._info_dict(info[0])

    def _reward_value(self, r):
        return int(r)

    def _info_dict(self, row):
        keys = getattr(self.batch, "info_keys", None) or RMSA_INFO_KEYS
        return {k: float(v) for k, v in zip(keys, row)}

    def get_number_slots(self, path, modulation=None):
        """rmsa_env.py:610-621 / rmcsa_env.py:753-765 (guard band included)."""
        import math

        mod = modulation if modulation is not None else path.best_modulation
        return math.ceil(self.current_service.bit_rate / (mod.spectral_efficiency * self.channel_width)) + 1

    def render(self, mode="human"):
        return

    def seed(self, seed=None):
        raise NotImplementedError("per-env seeds are fixed at construction (each env = reference env built with seed=...)")

    def close(self):
        if hasattr(self.batch, "close"):
            self.batch.close()

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


class RMCSAEnv(_SingleEnv):
    BATCH_CLS = BatchedRMCSAEnv

    def _make_spaces(self, kwargs):
        self.num_spatial_resources = kwargs.get("num_spatial_resources", 7)
        n_mod = len(self.topo.modulations)
        self.action_space = spaces.MultiDiscrete((self.k_paths + self.reject_action, n_mod,
                                                  self.num_spatial_resources + self.reject_action,
                                                  self.num_spectrum_resources + self.reject_action))
        self.observation_space = spaces.Dict({"topology": spaces.Discrete(10), "current_service": spaces.Discrete(10)})

    def _decode(self, a):
        return (int(a[0]), int(a[1]), int(a[2]), int(a[3]))


# ---- module-level heuristics with the reference's names ---------------------------------------------------
def shortest_path_first_fit(env):
    """rmsa_env.py:747-764 / deeprmsa_env.py:135-143 / rwa_env.py:425-435"""
    return env.unwrapped.policy_action("SP_FF")


def shortest_available_path_first_fit(env):
    """rmsa_env.py:767-779 / deeprmsa_env.py:146-155 / rwa_env.py:438-457"""
    return env.unwrapped.policy_action("SAP_FF")


def least_loaded_path_first_fit(env):
    """rmsa_env.py:782-803 / rwa_env.py:482-502"""
    return env.unwrapped.policy_action("LLP_FF")


def shortest_available_path_last_fit(env):
    """rwa_env.py:460-479"""
    return env.unwrapped.policy_action("SAP_LF")


def shortest_available_path_best_modulation_first_core_first_fit(env):
    """rmcsa_env.py:882-911"""
    return env.unwrapped.policy_action("SAP_BM_FC_FF")


def random_policy(env):
    """utils.py:99-100 (the stream is this package's numpy generator, not gym 0.21's)"""
    return env.action_space.sample()


def start_environment(env, steps):
    """utils.py:62-70"""
    done = True
    for _ in range(steps):
        if done:
            env.reset()
        while not done:
            _, _, done, _ = env.step(env.action_space.sample())
    return env


def evaluate_heuristic(env, heuristic, n_eval_episodes=10, render=False, callback=None, reward_threshold=None,
                       return_episode_rewards=False):
    """utils.py:103-141, same episode accounting: reset() (soft), loop until done, sum rewards."""
    episode_rewards, episode_lengths = [], []
    for _ in range(n_eval_episodes):
        _ = env.reset()
        done = False
        episode_reward = 0.0
        episode_length = 0
        while not done:
            action = heuristic(env)
            _, reward, done, _ = env.step(action)
            episode_reward += reward
            if callback is not None:
                callback(locals(), globals())
            episode_length += 1
            if render:
                env.render()
        episode_rewards.append(episode_reward)
        episode_lengths.append(episode_length)
    mean_reward = np.mean(episode_rewards)
    std_reward = np.std(episode_rewards)
    if reward_threshold is not None:
        assert mean_reward > reward_threshold, "Mean reward below threshold: {:.2f} < {:.2f}".format(mean_reward, reward_threshold)
    if return_episode_rewards:
        return episode_rewards, episode_lengths
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
    """rmsa_env.py:806-837, rmcsa_env.py:914-947: [one-hot(min(src,dst)), one-hot(max(src,dst)), slot map]."""

    def __init__(self, env):
        super().__init__(env)
        u = env.unwrapped
        cores = getattr(u, "num_spatial_resources", 1)
        shape = u.topo.n_nodes * 2 + u.topo.n_links * u.num_spectrum_resources * cores
        self.observation_space = spaces.Box(low=0, high=1, dtype=np.uint8, shape=(shape,))
        self.action_space = env.action_space

    def observation(self, observation=None):
        u = self.env.unwrapped
        n = u.topo.n_nodes
        svc = u.current_service
        tau = np.zeros((2, n))
        tau[0, min(svc.source_id, svc.destination_id)] = 1
        tau[1, max(svc.source_id, svc.destination_id)] = 1
        spectrum = np.asarray(u.topology.graph["available_slots"])
        return np.concatenate((tau.reshape(1, -1), spectrum.reshape(1, -1)), axis=1).reshape(self.observation_space.shape)

    def reset(self, **kw):
        self.env.reset(**kw)
        return self.observation()

    def step(self, action):
        _, r, d, i = self.env.step(action)
        return self.observation(), r, d, i


class PathOnlyFirstFitAction(_Wrapper):
    """rmsa_env.py:840-874 / rwa_env.py:505-536: the agent picks the path, first-fit picks the slot."""

    def __init__(self, env):
        super().__init__(env)
        u = env.unwrapped
        self.action_space = spaces.Discrete(u.k_paths + u.reject_action)
        self.observation_space = env.observation_space

    def action(self, action):
        u = self.env.unwrapped
        reject = (u.k_paths, u.num_spectrum_resources)
        if action >= u.k_paths:
            return reject
        svc = u.current_service
        path = u.k_shortest_paths[svc.source, svc.destination][action]
        avail = np.asarray(u.topology.graph["available_slots"])
        links = [int(x) for x in u.topo.path_links[svc.source_id, svc.destination_id, action][: path.hops]]
        free = np.all(avail[links, :] == 1, axis=0)
        S = u.num_spectrum_resources
        if isinstance(u, RWAEnv):
            idx = np.flatnonzero(free)  # rwa_env.py:524-532: all wavelengths are tried
            return (action, int(idx[0])) if len(idx) else reject
        n = u.get_number_slots(path)
        for s0 in range(0, S - int(n)):  # rmsa_env.py:856-858: same off-by-one as the heuristics
            if free[s0:s0 + int(n)].all():
                return (action, s0)
        return reject

    def step(self, action):
        return self.env.step(self.action(action))
