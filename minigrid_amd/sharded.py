"""ShardedVecEnv — one lockstep batch spread over the GPUs of a node, one process per GPU.

Environments never interact (the reference's `MiniGridEnv` instances share nothing on this path), so the batch
shards with NO data-path collective: rank g owns the contiguous global env indices [lo_g, hi_g) and seeds env i of
the whole batch with `seed + i` exactly like a single-process batch would (`gymnasium.vector.VectorEnv.reset(seed=int)`
semantics), which makes every result independent of the number of ranks.  The only exchange is optional: an
all-gather of the per-step outputs when ONE consumer needs the whole batch on every rank (RCCL over xGMI under
`torch.distributed` backend "nccl"; the same code runs on the "gloo" backend with CPU tensors in the test-suite).

    dist.init_process_group("nccl")                      # one process per GPU, torch.distributed.run / torchrun
    envs = ShardedVecEnv("MiniGrid-LavaCrossingS9N1-v0", 1_048_576, obs_mode="full", gather=True)
    obs, info = envs.reset(seed=0)                        # obs["image"]: (1_048_576, 9, 9, 3) on every rank
    obs, rew, term, trunc, info = envs.step(actions)     # `actions`: global (N,) or this rank's (hi-lo,) slice
"""
from __future__ import annotations

from typing import Any, Callable, Optional, Tuple

import numpy as np


def shard_range(num_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block partition of [0, num_envs): the first `num_envs % world_size` ranks get one extra env."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    if num_envs < world_size:
        raise ValueError(f"num_envs={num_envs} < world_size={world_size}: every rank needs at least one env")
    base, extra = divmod(num_envs, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _to_tensor(x):
    import torch
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x))


class ShardedVecEnv:
    """The `MiniGridVecEnv` surface for a batch that lives on `world_size` GPUs.

    `make` builds the local shard (default: `minigrid_amd.make_vec`, i.e. the HIP path; the CPU test-suite injects an
    oracle-backed stand-in to exercise the sharding/gather logic under gloo)."""

    def __init__(self, env_id: str, num_envs: int, *, gather: bool = True, group=None,
                 make: Optional[Callable[..., Any]] = None, **kwargs):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("ShardedVecEnv needs torch.distributed to be initialised (one process per GPU)")
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.num_envs = int(num_envs)
        self.gather = bool(gather)
        self.lo, self.hi = shard_range(self.num_envs, self.rank, self.world_size)
        self.local_num_envs = self.hi - self.lo
        self._max_local = -(-self.num_envs // self.world_size)
        if make is None:
            from .vector_env import make_vec as make
            kwargs.setdefault("output", "torch")
        self.local = make(env_id, self.local_num_envs, env_index_base=self.lo, **kwargs)
        self._missions = np.asarray(getattr(self.local, "_missions", ()))

    # ---- the one collective on the path -------------------------------------------------------------------
    def all_gather(self, x):
        """(local_n, ...) -> (num_envs, ...) on every rank.  Equal shards: one all_gather_into_tensor straight into
        the result; ragged shards are padded to the largest shard and trimmed."""
        import torch
        x = _to_tensor(x).contiguous()
        if self.world_size == 1:
            return x
        tail = tuple(x.shape[1:])
        if self.num_envs % self.world_size == 0:
            out = torch.empty((self.num_envs,) + tail, dtype=x.dtype, device=x.device)
            self._dist.all_gather_into_tensor(out, x, group=self.group)
            return out
        pad = torch.zeros((self._max_local,) + tail, dtype=x.dtype, device=x.device)
        pad[: x.shape[0]] = x
        out = torch.empty((self.world_size * self._max_local,) + tail, dtype=x.dtype, device=x.device)
        self._dist.all_gather_into_tensor(out, pad, group=self.group)
        parts = []
        for r in range(self.world_size):
            lo, hi = shard_range(self.num_envs, r, self.world_size)
            parts.append(out[r * self._max_local: r * self._max_local + (hi - lo)])
        return torch.cat(parts, 0)

    def _gather_obs(self, obs):
        if not self.gather:
            return obs
        if isinstance(obs, dict):
            out = {}
            for k, v in obs.items():
                if k == "mission":          # strings do not travel: gather the ids, map back on every rank
                    index = {m: i for i, m in enumerate(self._missions.tolist())}
                    ids = np.fromiter((index[m] for m in np.asarray(v).tolist()), np.uint8, len(v))
                    out[k] = self._missions[self.all_gather(ids).cpu().numpy()]
                else:
                    out[k] = self.all_gather(v)
            return out
        return self.all_gather(obs)

    # ---- Gymnasium VectorEnv surface ----------------------------------------------------------------------
    def _local_slice(self, seq, what):
        n = len(seq)
        if n == self.num_envs:
            return seq[self.lo:self.hi]
        if n == self.local_num_envs:
            return seq
        raise ValueError(f"{what} must have {self.num_envs} (global) or {self.local_num_envs} (this rank) entries, got {n}")

    def reset(self, *, seed: Any = None, options: Optional[dict] = None):
        if seed is not None and not isinstance(seed, (int, np.integer)):
            seed = list(self._local_slice(list(seed), "seed"))
        if options and options.get("reset_mask") is not None:
            options = dict(options, reset_mask=np.asarray(self._local_slice(np.asarray(options["reset_mask"]), "reset_mask")))
        obs, info = self.local.reset(seed=seed, options=options)     # int seed: the shard adds its env_index_base
        return self._gather_obs(obs), info

    def step(self, actions):
        a = self._local_slice(actions, "actions")
        obs, rew, term, trunc, info = self.local.step(a)
        if self.gather:
            rew, term, trunc = self.all_gather(rew), self.all_gather(term), self.all_gather(trunc)
        return self._gather_obs(obs), rew, term, trunc, info

    def close(self):
        self.local.close()

    def __getattr__(self, name):           # spaces, max_steps, counters(), rollout(), ... come from the local shard
        return getattr(self.local, name)
