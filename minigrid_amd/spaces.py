"""Observation/action space descriptors.  Uses gymnasium.spaces when gymnasium is installed, otherwise small
duck-typed stand-ins with the same attributes (shape, dtype, n, spaces) so that code reading the spaces works."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is optional
    from gymnasium.spaces import Box, Dict, Discrete, MultiDiscrete, Text  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # gymnasium absent: minimal descriptors
    HAVE_GYMNASIUM = False

    class _Space:
        def __init__(self, shape=None, dtype=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)

        def __repr__(self):
            return f"{type(self).__name__}({self.shape}, {self.dtype})"

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            super().__init__(shape, dtype)
            self.low = np.full(self.shape, low, dtype=dtype)
            self.high = np.full(self.shape, high, dtype=dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class Discrete(_Space):
        def __init__(self, n, start=0):
            super().__init__((), np.int64)
            self.n, self.start = int(n), int(start)

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

    class MultiDiscrete(_Space):
        def __init__(self, nvec, dtype=np.int64):
            self.nvec = np.asarray(nvec, dtype=dtype)
            super().__init__(self.nvec.shape, dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    class Dict(_Space):
        def __init__(self, spaces=None, **kw):
            super().__init__(None, None)
            self.spaces = dict(spaces or {})
            self.spaces.update(kw)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

    class Text(_Space):
        def __init__(self, max_length=256):
            super().__init__((), None)
            self.max_length = max_length


class MissionSpace(Text if HAVE_GYMNASIUM else object):  # type: ignore[misc]
    """The set of mission strings an env id can emit (reference: core/mission.py MissionSpace, reduced to the
    finite table this path needs)."""

    def __init__(self, missions):
        if HAVE_GYMNASIUM:
            super().__init__(max_length=max(len(m) for m in missions))
        self.missions = tuple(missions)
        self.shape = ()
        self.dtype = None

    def contains(self, x):
        return x in self.missions

    def __repr__(self):
        return f"MissionSpace({list(self.missions)!r})"
