"""The reference's gym registry (optical_rl_gym/__init__.py:3-26): the five ids, same kwargs.

`make(env_id, **kwargs)` works without gym.  When gymnasium or gym is importable, `register_envs()` (called on import of
this module) also registers the ids there, so that `gym.make("RMSA-v0", topology=..., seed=..., **kwargs)` of a
reference script returns this package's single-env front end.  QoSConstrainedRA-v0 is registered too: upstream its
constructor raises (qos_constrained_ra.py:32-41); here it works as the class is written (optical_rl_gym_amd/qos.py)."""
from . import gym_api, qos
from .envs import ENV_CLASSES

SINGLE = {"RMSA-v0": gym_api.RMSAEnv, "DeepRMSA-v0": gym_api.DeepRMSAEnv, "RWA-v0": gym_api.RWAEnv,
          "RMCSA-v0": gym_api.RMCSAEnv, "QoSConstrainedRA-v0": qos.QoSConstrainedRA}
ENTRY_POINTS = {"RMSA-v0": "optical_rl_gym_amd.gym_api:RMSAEnv", "DeepRMSA-v0": "optical_rl_gym_amd.gym_api:DeepRMSAEnv",
                "RWA-v0": "optical_rl_gym_amd.gym_api:RWAEnv", "RMCSA-v0": "optical_rl_gym_amd.gym_api:RMCSAEnv",
                "QoSConstrainedRA-v0": "optical_rl_gym_amd.qos:QoSConstrainedRA"}


def make(env_id, num_envs=None, **kwargs):
    """`make("RMSA-v0", topology=..., seed=10, ...)` -> the gym.Env-shaped 1-env front end;
    `make("RMSA-v0", num_envs=4096, seeds=..., ...)` -> the batch."""
    if num_envs is None:
        return SINGLE[env_id](**kwargs)
    return ENV_CLASSES[env_id](num_envs=num_envs, **kwargs)


def register_envs():
    """Register the ids with gymnasium / gym when installed; returns the names of the registries that took them."""
    done = []
    for modname in ("gymnasium", "gym"):
        try:
            import importlib

            reg = importlib.import_module(modname + ".envs.registration")
        except Exception:  # not installed
            continue
        for env_id, entry in ENTRY_POINTS.items():
            try:
                reg.register(id=env_id, entry_point=entry)
            except Exception:  # already registered
                pass
        done.append(modname)
    return done


REGISTERED_WITH = register_envs()
