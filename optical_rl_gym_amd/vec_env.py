"""stable-baselines3 `VecEnv` over a batched env.

SB3 drives N env copies through `reset()`, `step_async(actions)`, `step_wait()` and expects a soft reset right after an
env reports done, with the pre-reset observation under infos[i]["terminal_observation"] (DummyVecEnv behaviour;
reference usage: examples/stable_baselines3/DeepRMSA.ipynb cells 224-302, where the env sits behind
`Monitor(..., info_keywords=("episode_service_blocking_rate", "episode_bit_rate_blocking_rate"))`).
Here the N copies are one HIP batch: the soft reset happens inside the step kernel (auto_reset), and the Monitor's
per-episode bookkeeping (r, l, t + info keywords) is done on the host from the returned arrays.

The class implements every abstract method of `stable_baselines3.common.vec_env.VecEnv` (reset, step_async, step_wait,
close, get_attr, set_attr, env_method, env_is_wrapped, seed — plus get_images / render) and carries `num_envs`,
`observation_space`, `action_space`, so `PPO("MlpPolicy", OpticalVecEnv(batch))` can be constructed.  The spaces are
gymnasium / gym spaces when one of them is importable (SB3 checks their types), else the descriptions of spaces.py.
stable-baselines3 itself is not a dependency: if it is importable the class registers as a virtual subclass.
For an agent on the same GPU, `device_obs()` / `device_tensors()` hand out zero-copy views (DLPack) of the batch's
device-resident arrays; `obs_dtype=np.float32` casts observations for float32 policies.
"""
import os
import time

import numpy as np

from . import spaces as _own_spaces


def _space_module():
    for name in ("gymnasium.spaces", "gym.spaces"):
        try:
            import importlib

            return importlib.import_module(name)
        except Exception:  # not installed in the build image
            continue
    return _own_spaces


def make_spaces(batch, obs_dtype=np.float64, mod=None):
    """(observation_space, action_space) of one env of `batch`, as the reference defines them: rmsa_env.py:138-151,
    deeprmsa_env.py:38-45, rwa_env.py:72-85, rmcsa_env.py:181-196."""
    sp = mod or _space_module()
    rej = 1 if batch.allow_rejection else 0
    k, S = batch.k_paths, batch.num_spectrum_resources
    fam = batch.ENV_TYPE
    if fam == 1:
        obs = sp.Box(low=-2**30, high=2**30, shape=(batch.obs_dim,), dtype=obs_dtype)
        act = sp.Discrete(k * batch.j + rej)
    else:
        obs = sp.Dict({"topology": sp.Discrete(10), "current_service": sp.Discrete(10)})
        if fam == 4:  # QoSConstrainedRA takes the path index only (qos_constrained_ra.py:69)
            act = sp.Discrete(k + rej)
        elif fam == 3:
            act = sp.MultiDiscrete((k + rej, len(batch.modulation_formats), batch.num_spatial_resources + rej, S + rej))
        else:
            act = sp.MultiDiscrete((k + rej, S + rej))
    return obs, act


# Shared by every env without episode information in a step: an empty dict that REFUSES writes (a wrapper or callback that tries
# to put a key into it gets a TypeError instead of leaking the key into all 65 536 envs) and that copies / pickles as a plain
# empty dict — copy.deepcopy(infos) and pickle, which callbacks that snapshot their locals and HerReplayBuffer do, work as on
# the infos of SB3's own VecEnvs.  An env that reports done gets a mapping of its own.
class _NoInfo(dict):
    __slots__ = ()

    def _ro(self, *a, **k):
        raise TypeError("the info of an env without episode information is shared and read-only")

    __setitem__ = __delitem__ = update = setdefault = pop = popitem = clear = __ior__ = _ro

    def copy(self):
        return {}

    def __copy__(self):
        return {}

    def __deepcopy__(self, memo):
        return {}

    def __reduce__(self):
        return (dict, ())


_NO_INFO = _NoInfo()


class _FinishedStep:
    """What the envs that finished an episode in one step hand to SB3, kept as arrays: returns, lengths, info keywords, the
    terminal observations — one device gather and a handful of numpy calls per step whatever their number."""
    __slots__ = ("rets", "lens", "t", "keys", "cols", "term", "pos")

    def __init__(self, idx, rets, lens, t, keys, cols, term):
        self.rets, self.lens, self.t, self.keys, self.cols, self.term = rets, lens, t, keys, cols, term
        self.pos = dict(zip(idx, range(len(idx))))  # env -> its row

    def row(self, n):
        row = {"r": self.rets[n], "l": self.lens[n], "t": self.t}
        for k, col in zip(self.keys, self.cols):
            row[k] = col[n]
        return row


class _EpisodeInfo(dict):
    """The info dict of an env that finished an episode — {"episode": {r, l, t, **info_keywords}, "terminal_observation": row} —
    built when it is first looked at: SB3 reads `episode` of every finished env once (Monitor statistics) and
    `terminal_observation` only when it bootstraps a truncated episode; 1 300 envs finish per step at cfg3's steady state, and
    building every dict up front cost 2 ms of Python per step, creating a fresh lazy object per finished env still 1 ms.  So
    every env has ONE such object for the life of the VecEnv (created the first time an env finishes); a step only puts the
    objects of the envs that finished into the infos list.  Like the list itself it is valid until the next step_wait(): looked
    at later it shows the env's entry of the then current step, or nothing when the env did not finish in it.  copy() /
    copy.deepcopy / pickle give an independent plain dict (what VecMonitor and the replay buffers keep)."""
    __slots__ = ("_owner", "_i", "_serial")

    def __init__(self, owner, i):
        dict.__init__(self)
        self._owner, self._i, self._serial = owner, i, -1

    def _fill(self):
        o = self._owner
        if self._serial != o._serial:
            self._serial = o._serial
            dict.clear(self)
            src = o._fin_src
            n = src.pos.get(self._i) if src is not None else None
            if n is not None:
                dict.__setitem__(self, "episode", src.row(n))
                dict.__setitem__(self, "terminal_observation", None if src.term is None else src.term[n])
        return self

    def __getitem__(self, k):
        return dict.__getitem__(self._fill(), k)

    def get(self, k, default=None):
        return dict.get(self._fill(), k, default)

    def __contains__(self, k):
        return dict.__contains__(self._fill(), k)

    def __iter__(self):
        return dict.__iter__(self._fill())

    def __len__(self):
        return dict.__len__(self._fill())

    def keys(self):
        return dict.keys(self._fill())

    def items(self):
        return dict.items(self._fill())

    def values(self):
        return dict.values(self._fill())

    def __setitem__(self, k, v):
        dict.__setitem__(self._fill(), k, v)

    def __repr__(self):
        return dict.__repr__(self._fill())

    def __eq__(self, other):
        return dict.__eq__(self._fill(), other)

    __hash__ = None

    def copy(self):
        return dict(self._fill())

    def __copy__(self):
        return dict(self._fill())

    def __deepcopy__(self, memo):
        import copy

        return copy.deepcopy(dict(self._fill()), memo)

    def __reduce__(self):
        return (dict, (dict(self._fill()),))


class EpisodeLog:
    """Monitor-style rows (r, l, t, **info_keywords) of every finished episode, kept as numpy blocks appended once per step; reads
    like a list of dicts (len, iteration, indexing, truth value).  Bounded: beyond `max_rows` the oldest blocks are dropped —
    or appended to `spill_path` (a Monitor CSV, see OpticalVecEnv.save_monitor_csv) when one is set — so that a 10^9-step run
    keeps a flat memory footprint (cfg3's steady state finishes 1 300 episodes per step)."""

    def __init__(self, keys, max_rows=1 << 20, spill_path=None):
        self.keys = tuple(keys)
        self.max_rows = int(max_rows)
        self.spill_path = spill_path
        self._blocks = []  # [k, 3 + len(keys)] float64
        self._n = 0
        self.total = 0     # episodes ever logged (dropped or spilled ones included)
        self._spill_header = False

    def append_block(self, rets, lens, t, cols):
        k = len(rets)
        if not k:
            return
        blk = np.empty((k, 3 + len(self.keys)))
        blk[:, 0] = rets
        blk[:, 1] = lens
        blk[:, 2] = t
        for j, c in enumerate(cols):
            blk[:, 3 + j] = c
        self._blocks.append(blk)
        self._n += k
        self.total += k
        while self._n > self.max_rows and len(self._blocks) > 1:
            old = self._blocks.pop(0)
            self._n -= len(old)
            if self.spill_path:
                self._spill(old)

    def _spill(self, blk):
        with open(self.spill_path, "a") as f:
            for r in blk:
                f.write(self._csv_row(r) + "\n")

    def _csv_row(self, r):
        return ",".join([str(float(r[0])), str(int(r[1])), str(float(r[2]))] + [str(float(x)) for x in r[3:]])

    def _row(self, r):
        d = {"r": float(r[0]), "l": int(r[1]), "t": float(r[2])}
        for k, x in zip(self.keys, r[3:]):
            d[k] = float(x)
        return d

    def __len__(self):
        return self._n

    def __bool__(self):
        return self._n > 0

    def __iter__(self):
        for blk in list(self._blocks):
            for r in blk:
                yield self._row(r)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return list(self)[i]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        for blk in self._blocks:
            if i < len(blk):
                return self._row(blk[i])
            i -= len(blk)
        raise IndexError(i)

    def array(self):
        """Every kept row as one [n, 3 + len(keys)] float64 array (columns r, l, t, then the info keywords)."""
        return np.concatenate(self._blocks) if self._blocks else np.empty((0, 3 + len(self.keys)))


class OpticalVecEnv:
    metadata = {"render_modes": []}
    render_mode = None

    def __init__(self, batch, info_keywords=("episode_service_blocking_rate", "episode_bit_rate_blocking_rate"),
                 obs_dtype=np.float64, observation="default", max_logged_episodes=1 << 20, monitor_spill_path=None,
                 rates_only_info=False):
        """observation: "default" (DeepRMSA: its 1-D vector; other families: None, as their Dict observation holds live
        objects) or "matrix" (SimpleMatrixObservation built on the device: uint8 [2N + C*E*S]).
        max_logged_episodes / monitor_spill_path: the bound of `episode_log` and where rows beyond it go (EpisodeLog).
        rates_only_info (opt-in; the reference's step() always fills every info entry, rmsa_env.py:228-264): when the info
        keywords are blocking rates only, the step kernel may skip the compactness entries and the two link means — then
        `info_array()` / `device_tensors()["info"]` hold NaN in those columns instead of stale values, and close() puts the
        batch back into the full mode."""
        self.batch = batch
        self.num_envs = batch.num_envs
        self.obs_dtype = np.dtype(obs_dtype)
        self.observation_mode = observation
        self.observation_space, self.action_space = make_spaces(batch, obs_dtype)
        if observation == "matrix":
            t = batch.topology
            dim = 2 * t.n_nodes + batch.num_spatial_resources * t.n_links * batch.num_spectrum_resources
            self.observation_space = _space_module().Box(low=0, high=1, shape=(dim,), dtype=np.uint8)
        self.info_keywords = tuple(k for k in info_keywords if k in batch.info_keys)
        self._kw_idx = [batch.info_keys.index(k) for k in self.info_keywords]
        # what SB3 reads of info is the keywords of finished envs: when those are blocking rates only, the step kernel skips the
        # compactness entries and the per-step read of every link record behind the two link means (orl_batch_set_info_mode)
        rates = ("service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate", "episode_bit_rate_blocking_rate")
        self._rates_only = bool(rates_only_info and hasattr(batch, "set_info_mode")
                                and all(k in rates or k.startswith("bit_rate_blocking_") for k in self.info_keywords))
        if self._rates_only:
            batch.set_info_mode(True)
        self._actions = None
        self._obs_ring, self._obs_turn = None, 0
        # (batches that take `obs_out` in step(): the HIP batches; the oracle stand-in of the CPU tests does not)
        import inspect

        self._direct_obs = "obs_out" in inspect.signature(batch.step).parameters
        # (... and leave info on the device, handing out the rows of the envs that finished an episode)
        self._sparse_info = "fetch_info" in inspect.signature(batch.step).parameters and hasattr(batch, "info_rows")
        self._async = hasattr(batch, "step_async") and self._sparse_info  # (the HIP batches)
        self._queued = False
        self._infos, self._infos_set = None, ()
        self._info_pool, self._fin_src, self._serial = None, None, 0
        self._timing = {} if os.environ.get("ORL_VEC_TIMING") else None
        self._ep_ret = np.zeros(self.num_envs)
        self._ep_len = np.zeros(self.num_envs, np.int64)
        self._t0 = time.time()
        self.episode_log = EpisodeLog(self.info_keywords, max_logged_episodes, monitor_spill_path)  # Monitor-style rows

    # ---- VecEnv API ----
    def reset(self):
        self._ep_ret[:] = 0
        self._ep_len[:] = 0
        return self._obs(self.batch.reset(full=False))

    def step_async(self, actions):
        """Queues the whole step on the batch's stream — actions to the device, the step kernel, reward / done (/ observation)
        back into page-locked arrays — and returns: what the caller does until step_wait() overlaps the device's work."""
        self._actions = np.asarray(actions)
        self._queued = False
        if self._async:
            direct = self._direct()
            self.batch.step_async(self._actions, auto_reset=True, obs_out=self._next_obs_buffer() if direct else None,
                                  fetch_info=not self._sparse_info)
            self._queued = True

    def _next_obs_buffer(self):
        """Observations go straight into one of THREE page-locked arrays of the requested dtype, used in turn: the array a
        step returns stays untouched for the next two steps — SB3 holds `_last_obs` across exactly one further step — and no
        28-MB copy / dtype pass runs on the host per step (float32 is cast on the device)."""
        if self._obs_ring is None:
            make = getattr(self.batch, "host_array", None)
            shape = (self.num_envs, self.batch.obs_dim)
            self._obs_ring = [make(shape, self.obs_dtype) if make else np.zeros(shape, self.obs_dtype) for _ in range(3)]
        self._obs_turn = (self._obs_turn + 1) % 3
        return self._obs_ring[self._obs_turn]

    def _direct(self):
        return bool(self.observation_mode != "matrix" and getattr(self.batch, "obs_dim", 0)
                    and self.obs_dtype in (np.dtype(np.float64), np.dtype(np.float32)) and self._direct_obs)

    def step_wait(self):
        tm = self._timing  # (ORL_VEC_TIMING=1: seconds per section, tools/vec_env_episodes.py)
        if tm is not None:
            t_a = time.perf_counter()
        direct = self._direct()
        kw = dict(fetch_info=False) if self._sparse_info else {}
        if self._queued:
            self._queued = False
            obs, reward, done, info = self.batch.step_wait()
            if not direct:
                obs = self._obs(obs)
        elif direct:
            obs, reward, done, info = self.batch.step(self._actions, auto_reset=True, obs_out=self._next_obs_buffer(), **kw)
        else:
            obs, reward, done, info = self.batch.step(self._actions, auto_reset=True, **kw)
            obs = self._obs(obs)
        if tm is not None:
            t_b = time.perf_counter()
        self._ep_ret += reward
        self._ep_len += 1
        finished = np.flatnonzero(done)
        if tm is not None:
            t_c = time.perf_counter()
        if self._sparse_info:  # info stayed on the device: the rows of the envs that report done (row i of `rows` = finished[i])
            rows = self.batch.info_rows(finished)
            info = None
        if tm is not None:
            t_d = time.perf_counter()
        # SB3 wants one dict per env and only ever READS the ones of envs that did not finish an episode: those share one empty
        # read-only mapping (65 536 fresh dicts per step cost ten times the step itself); an env that reports done gets a dict of
        # its own.  The LIST is this object's, reused from step to step (a fresh 65 536-entry list per step costs as much as
        # the device's whole step): valid until the next step_wait(), like the arrays SB3's own VecEnvs hand out.
        infos = self._infos
        if infos is None:
            infos = self._infos = [_NO_INFO] * self.num_envs
        for i in self._infos_set:
            infos[i] = _NO_INFO
        idx = self._infos_set = finished.tolist()
        self._serial += 1
        self._fin_src = None
        if tm is not None:
            t_e = time.perf_counter()
        if idx:
            # (with 50-step episodes 1 300 of 65 536 envs finish per step: everything per env stays in numpy — one gather per
            # quantity — and each finished env gets a small mapping that builds its dict when SB3 looks at it)
            rets, lens = self._ep_ret[finished], self._ep_len[finished]
            cols = [np.array(rows[:, j] if info is None else info[finished, j]) for j in self._kw_idx]
            # the in-kernel reset is soft: the pending service (hence the observation) is unchanged by it
            term = None if obs is None else np.array(obs[finished])  # one copy; each env gets its row of it
            t_now = round(time.time() - self._t0, 6)
            src = self._fin_src = _FinishedStep(idx, rets.tolist(), lens.tolist(), t_now, self.info_keywords, [c.tolist() for c in cols], term)
            if tm is not None:
                t_f = time.perf_counter()
                tm["gather"] = tm.get("gather", 0.0) + t_f - t_e
            pool = self._info_pool
            if pool is None:  # (one object per env, created the first time any env finishes)
                pool = self._info_pool = [_EpisodeInfo(self, i) for i in range(self.num_envs)]
            for i in idx:
                infos[i] = pool[i]
            self.episode_log.append_block(rets, lens, t_now, cols)
            self._ep_ret[finished] = 0
            self._ep_len[finished] = 0
            if tm is not None:
                tm["objects+log"] = tm.get("objects+log", 0.0) + time.perf_counter() - t_f
        out = obs, np.array(reward), np.array(done, dtype=bool), infos
        if tm is not None:
            t_z = time.perf_counter()
            for k, v in (("wait", t_b - t_a), ("returns", t_c - t_b), ("info_rows", t_d - t_c), ("clear", t_e - t_d), ("total", t_z - t_a)):
                tm[k] = tm.get(k, 0.0) + v
            tm["steps"] = tm.get("steps", 0) + 1
        return out

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if self._rates_only:  # (the batch object is the caller's: it goes back as it came)
            self._rates_only = False
            try:
                self.batch.set_info_mode(False)
            except Exception:
                pass
        self.batch.close()

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def get_attr(self, attr_name, indices=None):
        """Per-env value of a public attribute of the reference env: counters, current_time, current_service fields,
        or any attribute the whole batch shares (k_paths, episode_length, ...)."""
        from .envs import COUNTER_NAMES

        idx = self._indices(indices)
        if attr_name in COUNTER_NAMES:
            c = self.batch.counters()[:, COUNTER_NAMES.index(attr_name)]
            return [int(c[i]) for i in idx]
        if attr_name == "current_time":
            return [float(self.batch.net_stats(i)[3]) for i in idx]
        if attr_name == "current_service":
            svc = self.batch.services()
            keys = ("arrival_time", "holding_time", "source_id", "destination_id", "bit_rate", "service_id")
            return [dict(zip(keys, svc[i])) for i in idx]
        if attr_name in ("render_mode",):
            return [None for _ in idx]
        if hasattr(self, "_extra_attrs") and attr_name in self._extra_attrs:
            return [self._extra_attrs[attr_name][i] for i in idx]
        return [getattr(self.batch, attr_name) for _ in idx]

    def set_attr(self, attr_name, value, indices=None):
        """Attributes set through the VecEnv live beside the batch (one value per env): the simulation parameters of a batch
        are fixed at construction, like a reference env's after __init__."""
        if not hasattr(self, "_extra_attrs"):
            self._extra_attrs = {}
        col = self._extra_attrs.setdefault(attr_name, [None] * self.num_envs)
        for i in self._indices(indices):
            col[i] = value

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        """Call a method of the reference env on the selected envs: reset / seed / observation / render are mapped onto the
        batch (masked where the method changes state); anything else is looked up on the batch and called once."""
        idx = self._indices(indices)
        mask = np.zeros(self.num_envs, np.uint8)
        mask[idx] = 1
        if method_name == "reset":
            only = method_kwargs.get("only_episode_counters", method_kwargs.get("only_counters", method_args[0] if method_args else True))
            obs = self.batch.reset(full=not only, mask=mask)
            return [None if obs is None else np.array(obs[i]) for i in idx]
        if method_name == "seed":
            s = method_args[0] if method_args else method_kwargs.get("seed")
            return self.seed(s, indices=idx)
        if method_name == "observation":
            obs = self._obs(self.batch.observation() if self.batch.obs_dim else None)
            return [None if obs is None else np.array(obs[i]) for i in idx]
        if method_name == "render":
            return [None for _ in idx]
        out = getattr(self.batch, method_name)(*method_args, **method_kwargs)
        return [out for _ in idx]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def seed(self, seed=None, indices=None):
        """VecEnv.seed: env i gets seed + i (optical_network_env.py:205-210 per env)."""
        idx = self._indices(indices)
        base = 41 if seed is None else int(seed)
        seeds = [base + i for i in range(self.num_envs)]
        mask = np.zeros(self.num_envs, np.uint8)
        mask[idx] = 1
        self.batch.seed(seeds, mask=mask)
        return [seeds[i] for i in idx]

    def get_images(self):
        return [None] * self.num_envs  # the reference's render() is a no-op (rmsa_env.py:361-362)

    def render(self, mode=None):
        return None

    @property
    def unwrapped(self):
        return self

    # ---- Monitor file ----
    def save_monitor_csv(self, path, env_id=None):
        """The episode log in stable-baselines3's Monitor file format — a JSON header line, then `r,l,t` plus the info
        keywords (cf. the reference's examples/heuristics/bkp/rmsa-heu/sap_ff.monitor.csv) — so that SB3's
        `load_results` / plotting read a batched run like a DummyVecEnv one."""
        import json

        with open(path, "w") as f:
            f.write("#%s\n" % json.dumps({"t_start": self._t0, "env_id": env_id or type(self.batch).__name__}))
            cols = ("r", "l", "t") + self.info_keywords
            f.write(",".join(cols) + "\n")
            for row in self.episode_log:
                f.write(",".join(str(row[c]) for c in cols) + "\n")
        # (rows beyond the log's bound went to monitor_spill_path as they were dropped, in the same column order)

    def info_array(self):
        """Last step's info as [num_envs, len(info_keys)] (cheaper than per-env dicts for large batches)."""
        return self.batch._info

    # ---- zero-copy views for an agent on the same GPU ----
    def device_tensors(self):
        """{"actions", "reward", "done", "info"[, "obs", "terminal_obs"]}: torch views of the batch's device arrays (DLPack /
        __cuda_array_interface__, no copy).  Write actions, `batch.step(None, auto_reset=True, fetch=False)`, `batch.sync()`."""
        names = ["actions", "reward", "done", "info"] + (["obs", "terminal_obs"] if self.batch.obs_dim else [])
        return {n: self.batch.device_tensor(n) for n in names}

    def _obs(self, obs):
        if self.observation_mode == "matrix":
            return self.batch.matrix_observation()
        if obs is None:
            return None
        return np.array(obs, dtype=self.obs_dtype)


try:  # optional: make isinstance(x, VecEnv) true without inheriting SB3's constructor requirements
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv  # type: ignore

    _SB3VecEnv.register(OpticalVecEnv)
except Exception:  # stable-baselines3 is not installed in the build image
    pass
