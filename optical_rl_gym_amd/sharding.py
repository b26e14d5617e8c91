"""Multi-GPU: contiguous blocks of env indices per device, no data-path collective.

Envs are independent, so env i's trajectory depends only on seeds[i] and its action stream: the number of GPUs (and how
the batch is cut) never changes a result.  Two ways to use several GPUs of a node:

  * one process per GPU (bench.py --gpus N under torchrun): rank r builds the batch for global env indices
    [lo, hi) = shard_range(total, r, world) with seeds base + index;
  * one process, `MultiDeviceBatch(..., device_ids=[0, 1, ...])`: one batch per device behind the interface of a single
    batch — actions are scattered and rewards / dones / infos / observations gathered per shard, each shard driven by its
    own host thread on its own stream (the library calls release the GIL), so one VecEnv / one agent process can drive all
    GPUs.  The SURVEY's `n_devices, device_ids` arguments of the batch constructor live here, above the C ABI, because a
    handle of the ABI is bound to one device.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def shard_range(n_envs_total, rank, world):
    """Contiguous, balanced partition of range(n_envs_total)."""
    per, rem = divmod(n_envs_total, world)
    lo = rank * per + min(rank, rem)
    hi = lo + per + (1 if rank < rem else 0)
    return lo, hi


def shard_seeds(base_seed, n_envs_total, rank, world):
    lo, hi = shard_range(n_envs_total, rank, world)
    return [base_seed + i for i in range(lo, hi)]


class MultiRunStats:
    """RunStats of a run over several shards (see MultiDeviceBatch.run)."""

    def __init__(self, parts):
        self.shards = list(parts)
        get = lambda p, name: getattr(p, name, 0) or 0  # (shards need not be HIP batches: from_shards takes any batch object)
        self.ms_total = max(float(get(p, "ms_total")) for p in parts)
        self.ms_policy = max(float(get(p, "ms_policy")) for p in parts)
        self.ms_step = max(float(get(p, "ms_step")) for p in parts)
        self.launches = sum(int(get(p, "launches")) for p in parts)
        self.n_kernels = int(get(parts[0], "n_kernels"))

    def kernels(self):
        if not hasattr(self.shards[0], "kernels"):
            return []
        first = self.shards[0].kernels()
        return [(name, max(p.kernels()[i][1] for p in self.shards)) for i, (name, _) in enumerate(first)]


class MultiDeviceBatch:
    """`num_envs` envs spread over `device_ids` (contiguous shards), with the methods of `BatchedOpticalEnv`."""

    def __init__(self, env_id, num_envs, seeds=None, device_ids=(0,), **kwargs):
        from .envs import ENV_CLASSES

        if seeds is None:
            seeds = [None] * num_envs
        elif np.isscalar(seeds):
            seeds = [int(seeds) + i for i in range(num_envs)]
        shards = []
        for r, dev in enumerate(device_ids):
            lo, hi = shard_range(num_envs, r, len(device_ids))
            shards.append(ENV_CLASSES[env_id](num_envs=hi - lo, seeds=list(seeds[lo:hi]), device_id=int(dev), **kwargs))
        self._init_from(shards)

    @classmethod
    def from_shards(cls, shards):
        """Wrap existing batches (any objects with the batch interface), in env-index order."""
        self = cls.__new__(cls)
        self._init_from(list(shards))
        return self

    def _init_from(self, shards):
        self.shards = shards
        self.bounds = np.cumsum([0] + [s.num_envs for s in shards])
        self.num_envs = int(self.bounds[-1])
        self._pool = ThreadPoolExecutor(max_workers=len(shards))
        first = shards[0]
        for name in ("topology", "info_keys", "obs_dim", "n_info", "ENV_TYPE", "N_ACTION", "episode_length", "allow_rejection",
                     "k_paths", "num_spectrum_resources", "num_spatial_resources", "j", "modulation_formats", "reject_action"):
            if hasattr(first, name):
                setattr(self, name, getattr(first, name))

    # ---- helpers ----
    def _map(self, fn):
        return list(self._pool.map(fn, range(len(self.shards))))

    def _cut(self, arr, r):
        return None if arr is None else np.asarray(arr)[self.bounds[r]:self.bounds[r + 1]]

    def _owner(self, env):
        r = int(np.searchsorted(self.bounds, env, side="right") - 1)
        return self.shards[r], int(env - self.bounds[r])

    @staticmethod
    def _cat(parts):
        return None if parts[0] is None else np.concatenate(parts, axis=0)

    # ---- the batch interface ----
    def reset(self, full=False, mask=None):
        return self._cat(self._map(lambda r: self.shards[r].reset(full=full, mask=self._cut(mask, r))))

    def seed(self, seeds, mask=None):
        if np.isscalar(seeds):
            seeds = [int(seeds) + i for i in range(self.num_envs)]
        self._map(lambda r: self.shards[r].seed(list(seeds[self.bounds[r]:self.bounds[r + 1]]), mask=self._cut(mask, r)))

    def set_info_mode(self, rates_only):
        # (shards without the switch — the oracle stand-in of the CPU tests — always write every entry)
        self._map(lambda r: self.shards[r].set_info_mode(rates_only) if hasattr(self.shards[r], "set_info_mode") else None)

    def set_paths(self, paths):
        self._map(lambda r: self.shards[r].set_paths(self._cut(paths, r)))

    def policy(self, policy, fetch=True, paths=None):
        out = self._map(lambda r: self.shards[r].policy(policy, fetch=fetch, paths=self._cut(paths, r)))
        return self._cat(out) if fetch else None

    def step(self, actions, auto_reset=False, fetch=True, obs_out=None, fetch_info=True):
        """As a single batch's step().  `obs_out` (a caller-owned [num_envs, obs_dim] array: every shard writes its rows) and
        `fetch_info=False` (info stays on the devices, `info_rows` reads what is needed) are handed to shards that take them —
        the HIP batches — and emulated for others (the oracle stand-in of the CPU tests)."""
        import inspect

        for r, sh in enumerate(self.shards):  # (refused before anything is modified on any shard, like a single batch's step)
            if hasattr(sh, "validate_actions"):
                sh.validate_actions(self._cut(actions, r))

        def one(r):
            sh = self.shards[r]
            params = inspect.signature(sh.step).parameters
            kw = {}
            if fetch and obs_out is not None and "obs_out" in params:
                kw["obs_out"] = obs_out[self.bounds[r]:self.bounds[r + 1]]
            if fetch and not fetch_info and "fetch_info" in params:
                kw["fetch_info"] = False
            out = sh.step(self._cut(actions, r), auto_reset=auto_reset, fetch=fetch, **kw)
            if fetch and obs_out is not None and "obs_out" not in kw and out[0] is not None:
                obs_out[self.bounds[r]:self.bounds[r + 1]] = out[0]
            return out

        out = self._map(one)
        if not fetch:
            return None
        infos = [o[3] for o in out]
        self._info_parts = infos  # (a shard that ignored fetch_info=False returned its array: info_rows reads from it)
        self._info = None if any(i is None for i in infos) else self._cat(infos)
        obs = obs_out if (obs_out is not None and getattr(self, "obs_dim", 0)) else self._cat([o[0] for o in out])
        return obs, self._cat([o[1] for o in out]), self._cat([o[2] for o in out]), (self._info if fetch_info else None)

    def policy_step(self, policy, auto_reset=False, fetch=True, paths=None):
        """policy() and step() on its actions in one launch per shard (BatchedOpticalEnv.policy_step)."""
        out = self._map(lambda r: self.shards[r].policy_step(policy, auto_reset=auto_reset, fetch=fetch, paths=self._cut(paths, r)))
        if not fetch:
            return None
        return tuple(self._cat([o[k] for o in out]) for k in range(5))

    def step_async(self, actions, auto_reset=False, obs_out=None, fetch_info=True):
        """First half of step() on every shard (`BatchedOpticalEnv.step_async`): each device gets its slice of the actions and its
        step queued on its own stream, one shard after the other from this thread — the calls only queue work, so all devices run
        at once without host threads — and `step_wait()` collects them.  Shards without the two halves (the oracle stand-in of the
        CPU tests) step synchronously here."""
        # every shard's slice is checked BEFORE a step is queued on any of them: an action outside the action space is refused
        # with nothing modified anywhere (include/orl.h), not with shards 0..r-1 stepped and their steps left pending
        for r, sh in enumerate(self.shards):
            if hasattr(sh, "validate_actions"):
                try:
                    sh.validate_actions(self._cut(actions, r))
                except IndexError as exc:
                    raise IndexError("shard %d (envs %d..%d): %s" % (r, self.bounds[r], self.bounds[r + 1] - 1, exc)) from None
        self._async_obs = obs_out
        self._async_sync = {}
        queued = []
        try:
            for r, sh in enumerate(self.shards):
                a = self._cut(actions, r)
                oo = None if obs_out is None else obs_out[self.bounds[r]:self.bounds[r + 1]]
                if hasattr(sh, "step_async"):
                    sh.step_async(a, auto_reset=auto_reset, obs_out=oo, fetch_info=fetch_info)
                    queued.append(sh)
                else:
                    out = sh.step(a, auto_reset=auto_reset)
                    if oo is not None and out[0] is not None:
                        oo[:] = out[0]
                    self._async_sync[r] = out
        except Exception:
            # (a device error on one shard: the steps already queued are collected, so that no shard is left with a pending step;
            # the batch is no longer consistent — the earlier shards are one step ahead — and the error says so)
            for sh in queued:
                try:
                    sh.step_wait()
                except Exception:
                    pass
            raise

    def step_wait(self):
        out = [self._async_sync[r] if r in self._async_sync else sh.step_wait() for r, sh in enumerate(self.shards)]
        infos = [o[3] for o in out]
        self._info_parts = infos
        self._info = None if any(i is None for i in infos) else self._cat(infos)
        obs = self._async_obs if (self._async_obs is not None and getattr(self, "obs_dim", 0)) else self._cat([o[0] for o in out])
        return obs, self._cat([o[1] for o in out]), self._cat([o[2] for o in out]), self._info

    def info_rows(self, indices):
        """Rows `indices` of the info arrays the last step left on the devices, in the order asked for."""
        idx = np.asarray(indices, np.int64).reshape(-1)
        out = np.empty((len(idx), self.n_info), np.float64)
        owner = np.searchsorted(self.bounds, idx, side="right") - 1
        for r in np.unique(owner):
            sel = np.flatnonzero(owner == r)
            local = idx[sel] - self.bounds[r]
            part = getattr(self, "_info_parts", [None] * len(self.shards))[r]
            out[sel] = part[local] if part is not None else self.shards[r].info_rows(local)
        return out

    def host_array(self, shape, dtype):
        """Page-locked host memory when the shards offer it (copies from every device then run at the full PCIe rate)."""
        first = self.shards[0]
        return first.host_array(shape, dtype) if hasattr(first, "host_array") else np.zeros(shape, dtype)

    def run(self, policy, n_steps, time_kernels=False):
        """Every shard runs its own device-resident loop concurrently.  Returns ONE stats object with the fields of a single
        batch's RunStats — the shards run side by side, so times are the MAX over shards and launches the sum — plus
        `.shards`, the per-shard RunStats in env-index order."""
        parts = self._map(lambda r: self.shards[r].run(policy, n_steps, time_kernels))
        return MultiRunStats(parts)

    def evaluate(self, policy, n_eval_episodes=10):
        out = self._map(lambda r: self.shards[r].evaluate(policy, n_eval_episodes))
        return self._cat([o[0] for o in out]), self._cat([o[1] for o in out])

    def observation(self):
        return self._cat(self._map(lambda r: self.shards[r].observation()))

    def matrix_observation(self):
        return self._cat(self._map(lambda r: self.shards[r].matrix_observation()))

    def sync(self):
        self._map(lambda r: self.shards[r].sync())

    def check(self):
        self._map(lambda r: self.shards[r].check())

    def counters(self):
        return self._cat(self._map(lambda r: self.shards[r].counters()))

    def services(self):
        return self._cat(self._map(lambda r: self.shards[r].services()))

    def active(self):
        return self._cat(self._map(lambda r: self.shards[r].active()))

    def flags(self):
        return self._cat(self._map(lambda r: self.shards[r].flags()))

    def totals(self):
        t = self._map(lambda r: self.shards[r].totals())
        return sum(x[0] for x in t), sum(x[1] for x in t)

    def slots(self, env=0):
        s, i = self._owner(env)
        return s.slots(i)

    def link_stats(self, env=0):
        s, i = self._owner(env)
        return s.link_stats(i)

    def net_stats(self, env=0):
        s, i = self._owner(env)
        return s.net_stats(i)

    def n_active(self, env=0):
        s, i = self._owner(env)
        return s.n_active(i)

    def action_histograms_of(self, env=0):
        s, i = self._owner(env)
        return s.action_histograms_of(i)

    def pending(self, env=0):
        s, i = self._owner(env)
        return s.pending(i)

    def device_tensor(self, name):
        """Per-shard zero-copy device tensors (one per GPU), in env-index order."""
        return [s.device_tensor(name) for s in self.shards]

    def close(self):
        for s in self.shards:
            s.close()
        self._pool.shutdown(wait=False)
