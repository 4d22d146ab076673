"""gymnasium.utils.seeding stand-in: np_random(seed) -> (Generator(PCG64(SeedSequence(seed))), seed)."""
from __future__ import annotations

import numpy as np

RandomNumberGenerator = RNG = np.random.Generator


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError(f"Seed must be a non-negative integer, got {seed!r}")
    seed_seq = np.random.SeedSequence(None if seed is None else int(seed))
    np_seed = seed_seq.entropy
    return np.random.Generator(np.random.PCG64(seed_seq)), np_seed
