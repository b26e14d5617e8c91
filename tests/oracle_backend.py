"""Adapter that lets the host-side mirrors (optical_rl_gym_amd.gym_api / vec_env) run on the CPU oracle in the
`-m "not gpu"` suite.  Test infrastructure only."""
import numpy as np

from optical_rl_gym_amd.envs import RMSA_INFO_KEYS
from optical_rl_gym_amd.topology import Topology
from oracle.oracle import OracleBatch


class OracleBackend(OracleBatch):
    def __init__(self, env_type, topology, seeds, **kw):
        super().__init__(env_type, topology, seeds, **kw)
        self.topology = Topology.load(topology)
        self.num_envs = self.n
        self.num_spatial_resources = self.C
        self.num_spectrum_resources = self.S
        rej = 1 if kw.get("allow_rejection", env_type in ("RWA", "QoSConstrainedRA")) else 0
        if env_type == "RWA":
            self.info_keys = (["service_blocking_rate", "episode_service_blocking_rate"]
                              + ["path_action_probability[%d]" % i for i in range(self.k + rej)]
                              + ["wavelength_action_probability[%d]" % i for i in range(self.S + rej)])
        elif env_type == "RMCSA":
            self.info_keys = RMSA_INFO_KEYS[:4]
        elif env_type == "QoSConstrainedRA":
            self.info_keys = RMSA_INFO_KEYS[:2]
        else:
            self.info_keys = list(RMSA_INFO_KEYS)
            if kw.get("bit_rate_selection") == "discrete":
                self.info_keys += ["bit_rate_blocking_%s" % b for b in kw.get("bit_rates", (10, 40, 100))] + ["fairness"]
        self._info = np.zeros((self.n, self.n_info))
        self.ENV_TYPE = {"RMSA": 0, "DeepRMSA": 1, "RWA": 2, "RMCSA": 3, "QoSConstrainedRA": 4}[env_type]
        self.episode_length = kw.get("episode_length", 1000)
        self.k_paths = self.k
        self.allow_rejection = bool(rej)
        self.j = kw.get("j", 1)
        self.reject_action = rej
        self.modulation_formats = list(self.topology.modulations)

    def evaluate(self, policy, n_eval_episodes=10):
        """Same contract as BatchedOpticalEnv.evaluate, by stepping the oracle."""
        n = int(n_eval_episodes)
        L = self.episode_length if self.ENV_TYPE == 2 else self.episode_length - 1
        super().reset(full=False)
        rewards = np.zeros((self.n, n))
        for ep in range(n):
            for t in range(L):
                last = ep == n - 1 and t == L - 1
                _, r, d, _ = self.step(self.policy(policy), auto_reset=not last)
                rewards[:, ep] += r
            assert d.all()
        return rewards, np.full((self.n, n), L, np.int64)

    def step(self, actions, auto_reset=False, fetch=True):
        out = super().step(actions, auto_reset=auto_reset)
        self._info = out[3]
        return out

    def policy(self, policy, fetch=True, paths=None):
        return super().policy(policy, paths=paths)

    def run(self, policy, n_steps, time_kernels=False):
        return super().run(policy, n_steps)

    def reset(self, full=False, mask=None):
        super().reset(full=full, mask=mask)
        return self.observation() if self.obs_dim else None

    def close(self):
        pass
