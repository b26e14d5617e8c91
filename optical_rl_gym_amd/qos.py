"""QoSConstrainedRA: the gym.Env-shaped 1-env front end, its heuristics and its observation wrapper
(reference: optical_rl_gym/envs/qos_constrained_ra.py).

The reference class cannot be constructed as shipped (its __init__ hands a `k_paths` keyword to a base class that does not
take it, :32-41, and builds `Service(..., service_class=...)`, a field utils.Service does not have); what is reproduced
here is the class as written, pinned by fixtures captured from the reference with those two things repaired at import
time (oracle/gen_golden_qos.py).  Per link the env keeps a counter of free spectrum units
(`topology.graph["available_spectrum"]`), services need one unit on every link of their path, class-0 services may only
take the shortest path, and the reward of an accepted service is its class's `classes_reward`."""
import numpy as np

from . import spaces
from .envs import BatchedQoSConstrainedRA
from .gym_api import Service, _device_heuristic, _SingleEnv, _Wrapper


class QoSConstrainedRA(_SingleEnv):
    BATCH_CLS = BatchedQoSConstrainedRA
    DEFAULT_SLOTS = 80
    DEFAULT_REJECTION = True
    metadata = {"metrics": ["service_blocking_rate", "episode_service_blocking_rate"]}

    def __init__(self, topology=None, seed=None, _backend=None, **kwargs):
        self.num_service_classes = kwargs.get("num_service_classes", 1)
        self.classes_arrival_probabilities = list(kwargs.get("classes_arrival_probabilities", [1.0]))
        self.classes_reward = list(kwargs.get("classes_reward", [1.0]))
        super().__init__(topology, seed, _backend, **kwargs)

    def _make_spaces(self, kwargs):
        self.action_space = spaces.Discrete(self.k_paths + self.reject_action)  # qos_constrained_ra.py:69
        self.observation_space = spaces.Dict({"topology": spaces.Discrete(10), "current_service": spaces.Discrete(10)})

    @property
    def service(self):
        at, ht, src, dst, clazz, sid = self.batch.services()[0]
        names = self.topo.node_names
        s = Service(int(sid), names[int(src)], int(src), names[int(dst)], int(dst), float(at), float(ht), None,
                    number_slots=1, accepted=self._accepted)
        s.service_class = int(clazz)
        return s

    current_service = service

    def reset(self, only_counters=True):
        self.batch.reset(full=not only_counters)
        return self.observation()

    def observation(self):
        return {"topology": self.topology, "service": self.service}

    def _encode(self, action):
        return np.array([[int(action)]], dtype=np.int64)

    def _decode(self, a):
        return int(a[0])

    def _reward_value(self, r):
        return float(r)

    def _is_path_free(self, path, number_slots):
        return is_path_free(self.topology, path, number_slots)


def _links(topology, path):
    env = topology._env
    return env._links(path)


def is_path_free(topology, path, number_slots):
    """qos_constrained_ra.py:381-392"""
    if number_slots > topology.graph["num_spectrum_resources"]:
        return False
    return bool(np.all(topology.graph["available_spectrum"][_links(topology, path)] >= number_slots))


def get_path_capacity(topology, path):
    """qos_constrained_ra.py:395-405"""
    return min([np.finfo(0.0).max] + [x for x in topology.graph["available_spectrum"][_links(topology, path)]])


@_device_heuristic("SP_FF")
def shortest_path(env):
    """qos_constrained_ra.py:408-415"""
    return env.unwrapped.policy_action("SP_FF")


@_device_heuristic("SAP_FF")
def shortest_available_path(env):
    """qos_constrained_ra.py:418-432"""
    return env.unwrapped.policy_action("SAP_FF")


@_device_heuristic("LLP_FF")
def least_loaded_path(env):
    """qos_constrained_ra.py:435-450"""
    return env.unwrapped.policy_action("LLP_FF")


class MatrixObservationWithPaths(_Wrapper):
    """qos_constrained_ra.py:453-513: per link the used units as a run of ones, then — per candidate path — the usage the
    link would have with the service on it; the service class at the end."""

    def __init__(self, env):
        super().__init__(env)
        u = env.unwrapped
        shape = u.topo.n_links * u.num_spectrum_resources * (u.k_paths + 1) + 1
        self.observation_space = spaces.Box(low=0, high=1, dtype=np.uint8, shape=(shape,))
        self.action_space = env.action_space

    def observation(self, observation=None):
        u = self.env.unwrapped
        S, E = u.num_spectrum_resources, u.topo.n_links
        avail = u.batch.spectrum(0).astype(int)
        obs = np.zeros((E, S * (u.k_paths + 1)))
        for link in range(E):
            obs[link, 0:S - avail[link]] = 1
        svc = u.service
        for idp, path in enumerate(u.k_shortest_paths[svc.source, svc.destination]):
            start = (idp + 1) * S
            for link in u._links(path):
                obs[link, start:start + S - avail[link] + 1] = 1
            if svc.service_class == 0:
                break  # high-priority services only accept the shortest path
        return np.concatenate((obs.reshape((1, -1)), np.array([[float(svc.service_class)]])), axis=1)

    def reset(self, **kw):
        self.env.reset(**kw)
        return self.observation()

    def step(self, action):
        _, r, d, i = self.env.step(action)
        return self.observation(), r, d, i
