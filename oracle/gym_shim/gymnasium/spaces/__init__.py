"""gymnasium.spaces stand-in: just enough of Space/Box/Discrete/Dict/MultiDiscrete."""
from __future__ import annotations

from typing import Any, Generic, TypeVar

import numpy as np

from ..utils import seeding

T_cov = TypeVar("T_cov", covariant=True)


class Space(Generic[T_cov]):
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None or dtype is str else np.dtype(dtype)
        self._np_random = None
        if seed is not None:
            if isinstance(seed, np.random.Generator):
                self._np_random = seed
            else:
                self.seed(seed)

    @property
    def shape(self):
        return self._shape

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    def seed(self, seed=None):
        self._np_random, np_seed = seeding.np_random(seed)
        return np_seed

    def sample(self, mask=None):
        raise NotImplementedError

    def contains(self, x) -> bool:
        raise NotImplementedError

    def __contains__(self, x) -> bool:
        return self.contains(x)


class Box(Space[np.ndarray]):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.asarray(low).shape
        self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        super().__init__(shape, dtype, seed)

    def sample(self, mask=None):
        if np.issubdtype(self.dtype, np.integer):
            return self.np_random.integers(self.low, self.high, endpoint=True, dtype=self.dtype)
        return self.np_random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(Space[np.int64]):
    def __init__(self, n, seed=None, start=0):
        self.n = int(n)
        self.start = int(start)
        super().__init__((), np.int64, seed)

    def sample(self, mask=None):
        return self.start + self.np_random.integers(self.n)

    def contains(self, x):
        try:
            return self.start <= int(x) < self.start + self.n
        except Exception:
            return False


class MultiDiscrete(Space[np.ndarray]):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.array(nvec, dtype=dtype, copy=True)
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self, mask=None):
        return (self.np_random.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))


class Dict(Space[dict]):
    def __init__(self, spaces=None, seed=None, **kw):
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)
        super().__init__(None, None, seed)

    def __getitem__(self, key):
        return self.spaces[key]

    def __setitem__(self, key, value):
        self.spaces[key] = value

    def keys(self):
        return self.spaces.keys()

    def sample(self, mask=None):
        return {k: s.sample() for k, s in self.spaces.items()}

    def contains(self, x):
        return isinstance(x, dict) and all(k in x and self.spaces[k].contains(x[k]) for k in self.spaces)


class Text(Space[str]):
    def __init__(self, max_length, seed=None, **kw):
        self.max_length = max_length
        super().__init__((), None, seed)
