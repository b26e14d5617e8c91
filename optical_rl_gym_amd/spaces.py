"""Minimal action/observation space descriptions (gym is not a dependency of the hot path).

Shapes and bounds follow the reference: rmsa_env.py:138-151, deeprmsa_env.py:38-45, rwa_env.py:72-85,
rmcsa_env.py:181-196.  `sample()` draws from a numpy generator; it is NOT gym 0.21's stream.
"""
import numpy as np


class Space:
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), dtype
        self._rng = np.random.default_rng(0)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(None if seed is None else int(seed) % (2**63))
        return [seed]


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = int(n)

    def sample(self):
        return int(self._rng.integers(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, np.int64)
        super().__init__(self.nvec.shape, np.int64)

    def sample(self):
        return self._rng.integers(self.nvec)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.nvec.shape and bool(((0 <= x) & (x < self.nvec)).all())


class Box(Space):
    def __init__(self, low, high, shape, dtype=np.float32):
        super().__init__(shape, dtype)
        self.low, self.high = low, high

    def sample(self):
        return self._rng.uniform(0, 1, self.shape).astype(self.dtype)


class Dict(Space):
    def __init__(self, spaces):
        super().__init__((), None)
        self.spaces = dict(spaces)

    def seed(self, seed=None):
        return [s.seed(seed) for s in self.spaces.values()]

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}
