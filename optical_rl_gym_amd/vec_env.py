"""stable-baselines3 `VecEnv`-shaped adapter over a batched env.

SB3 drives N env copies through `reset()`, `step_async(actions)`, `step_wait()` and expects a soft reset
right after an env reports done, with the pre-reset observation under infos[i]["terminal_observation"]
(DummyVecEnv behaviour; reference usage: examples/stable_baselines3/DeepRMSA.ipynb cells 224-302, where the
env sits behind `Monitor(..., info_keywords=("episode_service_blocking_rate", "episode_bit_rate_blocking_rate"))`).
Here the N copies are one HIP batch; the soft reset happens inside the step kernel (auto_reset), and the
Monitor's per-episode bookkeeping (r, l, t + info keywords) is done on the host from the returned arrays.
stable-baselines3 itself is not a dependency: if it is importable the class registers as a virtual subclass.
"""
import time

import numpy as np


class OpticalVecEnv:
    def __init__(self, batch, info_keywords=("episode_service_blocking_rate", "episode_bit_rate_blocking_rate")):
        self.batch = batch
        self.num_envs = batch.num_envs
        self.info_keywords = tuple(k for k in info_keywords if k in batch.info_keys)
        self._kw_idx = [batch.info_keys.index(k) for k in self.info_keywords]
        self._actions = None
        self._ep_ret = np.zeros(self.num_envs)
        self._ep_len = np.zeros(self.num_envs, np.int64)
        self._t0 = time.time()
        self.episode_log = []  # Monitor-style rows: dict(r, l, t, **info_keywords)

    # ---- VecEnv API ----
    def reset(self):
        self._ep_ret[:] = 0
        self._ep_len[:] = 0
        return self._obs(self.batch.reset(full=False))

    def step_async(self, actions):
        self._actions = np.asarray(actions)

    def step_wait(self):
        obs, reward, done, info = self.batch.step(self._actions, auto_reset=True)
        self._ep_ret += reward
        self._ep_len += 1
        infos = [{} for _ in range(self.num_envs)]
        for i in np.flatnonzero(done):
            row = dict(r=float(self._ep_ret[i]), l=int(self._ep_len[i]), t=round(time.time() - self._t0, 6))
            for k, j in zip(self.info_keywords, self._kw_idx):
                row[k] = float(info[i, j])
            infos[i]["episode"] = row
            # the in-kernel reset is soft: the pending service (hence the observation) is unchanged by it
            infos[i]["terminal_observation"] = None if obs is None else np.array(obs[i])
            self.episode_log.append(row)
            self._ep_ret[i] = 0
            self._ep_len[i] = 0
        return self._obs(obs), np.array(reward), np.array(done, dtype=bool), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

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

    def info_array(self):
        """Last step's info as [num_envs, len(info_keys)] (cheaper than per-env dicts for large batches)."""
        return self.batch._info

    def close(self):
        self.batch.close()

    def get_attr(self, name, indices=None):
        idx = range(self.num_envs) if indices is None else indices
        if name in ("services_processed", "services_accepted", "episode_services_processed",
                    "episode_services_accepted", "bit_rate_requested", "bit_rate_provisioned",
                    "episode_bit_rate_requested", "episode_bit_rate_provisioned"):
            from .envs import COUNTER_NAMES

            c = self.batch.counters()[:, COUNTER_NAMES.index(name)]
            return [int(c[i]) for i in idx]
        return [getattr(self.batch, name) for _ in idx]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * (self.num_envs if indices is None else len(indices))

    def seed(self, seed=None):
        return [None] * self.num_envs  # seeds are fixed at construction

    def _obs(self, obs):
        return None if obs is None else np.array(obs)


try:  # optional: make isinstance(x, VecEnv) true without inheriting SB3's constructor requirements
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv  # type: ignore

    _SB3VecEnv.register(OpticalVecEnv)
except Exception:  # stable-baselines3 is not installed in the build image
    pass
