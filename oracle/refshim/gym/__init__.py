"""Minimal stand-in for the `gym` 0.21 package — GOLDEN-GENERATION TOOLING ONLY.

`gym` is not installed in the build container (no network).  The reference
(/root/reference/optical_rl_gym) only needs gym's base classes, four space types and
the registry to be importable; none of that is on the step() path whose outputs we
capture.  oracle/gen_golden.py puts this directory on sys.path *ahead of* the
reference so `import optical_rl_gym` works.  Nothing in the product or in the tests
imports this module.  `Space.sample()` exists only so random_policy() does not crash;
its stream is NOT gym's and is never used for a golden vector.
"""
import importlib

from . import spaces  # noqa: F401
from .envs import registration as _registration


class Env:
    metadata = {}
    action_space = None
    observation_space = None
    reward_range = (-float("inf"), float("inf"))

    @property
    def unwrapped(self):
        return self

    def seed(self, seed=None):
        return [seed]

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.metadata = env.metadata

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)


class ObservationWrapper(Wrapper):
    def reset(self, **kwargs):
        return self.observation(self.env.reset(**kwargs))

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self.observation(obs), reward, done, info


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))


class RewardWrapper(Wrapper):
    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return obs, self.reward(reward), done, info


def make(env_id, **kwargs):
    entry = _registration.registry[env_id]
    module_name, cls_name = entry.split(":")
    cls = getattr(importlib.import_module(module_name), cls_name)
    return cls(**kwargs)
