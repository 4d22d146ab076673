"""Test helper: the CPU oracle over a large batch, sharded over a thread pool (ctypes drops the GIL inside liboracle.so), so that the
full BASELINE batch sizes can be replayed for hundreds of steps in seconds.  Env i of the batch is seeded seed0 + i like
`MiniGridVecEnv.reset(seed=seed0)`."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class ParOracle:
    def __init__(self, env_id, n, full=False, threads=None, **over):
        from oracle import oracle as O
        self.O = O
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        self.threads = max(1, min(threads or cores, 64, n // 256 or 1))
        edges = np.linspace(0, n, self.threads + 1).astype(int)
        self.ranges = [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]
        self.vecs = [O.OracleVec(env_id, b - a, full_obs=full, **over) for a, b in self.ranges]
        self.pool = ThreadPoolExecutor(len(self.vecs))
        self.n = n

    def _map(self, fn):
        return list(self.pool.map(fn, range(len(self.vecs))))

    def reset(self, seed0):
        outs = self._map(lambda k: self.vecs[k].reset(seeds=np.arange(seed0 + self.ranges[k][0], seed0 + self.ranges[k][1], dtype=np.uint64)))
        return tuple(np.concatenate([o[j] for o in outs]) for j in range(3))

    def step(self, actions):
        outs = self._map(lambda k: self.vecs[k].step(actions[self.ranges[k][0]:self.ranges[k][1]]))
        return tuple(np.concatenate([o[j] for o in outs]) for j in range(6))

    def step_quiet(self, actions):
        outs = self._map(lambda k: self.vecs[k].step_quiet(actions[self.ranges[k][0]:self.ranges[k][1]]))
        return tuple(np.concatenate([o[j] for o in outs]) for j in range(3))

    def philox_step(self, action_seed, t, quiet=True, env_base=0):
        """One step under the product's device policy (mg_rollout's Philox actions for step counter t)."""
        def one(k):
            a, b = self.ranges[k]
            act = self.O.philox_actions(action_seed, t, b - a, env_base=env_base + a)
            return (self.vecs[k].step_quiet(act) if quiet else self.vecs[k].step(act)) + (act,)
        outs = self._map(one)
        return tuple(np.concatenate([o[j] for o in outs]) for j in range(len(outs[0])))

    def get_state(self):
        outs = self._map(lambda k: self.vecs[k].get_state())
        return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])

    def get_rng(self):
        return np.concatenate(self._map(lambda k: self.vecs[k].get_rng()))

    def close(self):
        self.pool.shutdown()
