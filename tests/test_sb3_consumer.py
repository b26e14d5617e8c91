"""The real consumer (reference usage examples/stable_baselines3/DeepRMSA.ipynb cells 272-302:
Monitor -> DummyVecEnv -> PPO): stable-baselines3 itself driving `OpticalVecEnv`.  stable-baselines3 / gymnasium are not
part of the build image, so these tests skip there; in any image that has them they check what SB3's own isinstance /
getattr probes, `VecMonitor` and a short `PPO.learn` need from the class.  CPU: the oracle stands in for the batch (test
infrastructure); GPU: the HIP batch."""
import numpy as np
import pytest

sb3 = pytest.importorskip("stable_baselines3")

KW = dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=1.0 / 12.0, j=1, episode_length=12)


def _drive(batch):
    from stable_baselines3 import PPO
    from stable_baselines3.common.vec_env import VecEnv, VecMonitor

    from optical_rl_gym_amd.vec_env import OpticalVecEnv

    venv = OpticalVecEnv(batch, obs_dtype=np.float32)
    assert isinstance(venv, VecEnv)  # registered as a virtual subclass: SB3 does not wrap it into a DummyVecEnv
    mon = VecMonitor(venv, info_keywords=("episode_service_blocking_rate",))
    obs = mon.reset()
    assert obs.shape == (batch.num_envs, batch.obs_dim)
    n_done = 0
    for _ in range(30):
        acts = np.array([mon.action_space.sample() for _ in range(batch.num_envs)])
        obs, rew, done, infos = mon.step(acts)
        assert len(infos) == batch.num_envs
        for i in np.flatnonzero(done):
            ep = infos[i]["episode"]  # VecMonitor's own bookkeeping on top of ours
            assert ep["l"] == KW["episode_length"] - 1 and "episode_service_blocking_rate" in ep
            assert infos[i]["terminal_observation"].shape == (batch.obs_dim,)
            n_done += 1
    assert n_done >= batch.num_envs
    model = PPO("MlpPolicy", venv, n_steps=8, batch_size=8 * batch.num_envs, n_epochs=1, seed=3, device="cpu")
    model.learn(total_timesteps=64 * batch.num_envs // 8)
    assert model.num_timesteps >= 8 * batch.num_envs
    assert len(venv.episode_log) > 0


def test_sb3_drives_the_vec_env_on_the_oracle_stand_in():
    from tests.oracle_backend import OracleBackend

    _drive(OracleBackend("DeepRMSA", "nsfnet_chen", [3 + i for i in range(8)], **KW))


@pytest.mark.gpu
def test_sb3_drives_the_vec_env_on_the_hip_batch():
    import optical_rl_gym_amd as orl

    batch = orl.make("DeepRMSA", topology="nsfnet_chen", num_envs=16, seeds=[3 + i for i in range(16)], **KW)
    _drive(batch)
