"""Space stand-ins (shape/nvec/n attributes + a non-gym sample()); tooling only."""
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = shape
        self.dtype = dtype
        self._rng = np.random.RandomState(0)

    def seed(self, seed=None):
        self._rng = np.random.RandomState(None if seed is None else int(seed) % (2**32))
        return [seed]


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = n

    def sample(self):
        return int(self._rng.randint(self.n))


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        super().__init__(self.nvec.shape, np.int64)

    def sample(self):
        return (self._rng.random_sample(self.nvec.shape) * self.nvec).astype(np.int64)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        super().__init__(tuple(shape) if shape is not None else np.shape(low), dtype)
        self.low, self.high = low, high

    def sample(self):
        return self._rng.uniform(0, 1, self.shape).astype(self.dtype)


class Dict(Space):
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = spaces

    def seed(self, seed=None):
        return [s.seed(seed) for s in self.spaces.values()]

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}
