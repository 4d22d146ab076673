"""ShardedVecEnv — one lockstep batch spread over the GPUs of a node, one process per GPU.

Environments never interact (the reference's `MiniGridEnv` instances share nothing on this path), so the batch
shards with NO data-path collective: rank g owns the contiguous global env indices [lo_g, hi_g) and seeds env i of
the whole batch with `seed + i` exactly like a single-process batch would (`gymnasium.vector.VectorEnv.reset(seed=int)`
semantics), which makes every result independent of the number of ranks.  The only exchange is optional: an all-gather of the
step's outputs when one consumer needs the whole batch on every rank.  The step kernel writes a step's outputs as one
contiguous record {image (n, ...) | (n) x mg_step_scalars 16 B [| sentence (n, 2) u64]} (mg_outputs in include/minigrid_hip.h).
Per step the record's PARTS are gathered field-major -- one `all_gather_into_tensor` for the images, one for the 16-byte scalar
entries (one more for the sentence levels' mission words), each straight out of the record (zero-copy send) into a persistent
(world x per-rank) buffer -- so that the GLOBAL per-field tensors are views of the gathered buffers: the image is the image
buffer itself, reward / terminated / ... are strided views of the scalar buffer; nothing is re-assembled by a second pass
(round 5 gathered whole records, [rank][image | scalars], and concatenated every field again: 255 MB per step for
LavaCrossing FullyObs x 1 048 576 on 8 GPUs).  RCCL over xGMI under `torch.distributed` backend "nccl"; the same code runs on
the "gloo" backend with CPU tensors in the test-suite.  Ragged batches (N % world != 0): the short ranks send from a
persistent padded buffer (allocated once) and the global tensors are compacted (the only case with a copy).

Stream ordering, no host synchronisation: with `output="torch"` the shard's HIP stream is created BLOCKING when torch's
current stream is the legacy NULL stream (vector_env.py), i.e. NULL-stream work -- which is what RCCL's launch is ordered
against -- starts after the step kernel, and the next step kernel starts after the NULL stream has waited for the
collective.  A shard on a non-blocking private stream is synchronised explicitly instead.

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


SCALAR_STRIDE = 16          # sizeof(mg_step_scalars), include/minigrid_hip.h
_SCALAR_FIELD = {"reward": 0, "terminated": 8, "truncated": 9, "direction": 10, "action": 11, "mission_id": 12}   # offsetof(...)


def record_layout(n: int, obs_bytes: int, sentence: bool = False) -> dict:
    """Byte offsets of the fields of one step record for a shard of n envs -- the layout mg_create gives a trajectory slot
    (minigrid_amd/csrc/mg_api.hip; tests check it against mg_get_outputs): {image (n, ...) | (n) x mg_step_scalars [| sentence (n, 2) u64]},
    every part on a 256-byte boundary.  The scalar entries are 16 bytes per env: the offsets below are env 0's fields, env i's lie
    i * scalar_stride further."""
    up = lambda v: (v + 255) & ~255
    off = {"image": 0, "scalar_stride": SCALAR_STRIDE}
    base = up(n * obs_bytes + 16)
    for name, o in _SCALAR_FIELD.items():
        off[name] = base + o
    off["record_bytes"] = base + up(n * SCALAR_STRIDE)
    if sentence:                      # the sentence levels: the mission as data, two u64 per env (mg_outputs.sentence)
        off["sentence"] = off["record_bytes"]
        off["record_bytes"] = up(off["sentence"] + 16 * n)
    return off


def scalar_field(raw, name: str, n: int, lay: dict):
    """Zero-copy strided view of scalar field `name` of the n envs of one step record `raw` (1-D uint8 tensor of >= record_bytes)."""
    import torch
    base = lay["reward"]
    area = raw[base: base + n * SCALAR_STRIDE]
    if name == "reward":
        return area.view(torch.float64)[0::2]
    if name == "mission_id":
        return area.view(torch.int16)[6::8]
    return area.view(n, SCALAR_STRIDE)[:, _SCALAR_FIELD[name]]


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
        self._step_stream = self._comm_stream = None
        if make is None:
            from .vector_env import make_vec as make
            kwargs.setdefault("output", "torch")
            if self.gather and kwargs["output"] == "torch" and kwargs.get("stream") is None:
                # The shard steps on its OWN (non-blocking) stream and the collectives run on a second one, ordered by events
                # (rollout_gather): launch k + 1 runs under the gather of launch k.  Two blocks of max_fused_steps trajectory slots
                # rotate, hence 64 slots.
                import torch
                dev = kwargs.get("device")
                self._step_stream = torch.cuda.Stream(device=dev)
                self._comm_stream = torch.cuda.Stream(device=dev)
                kwargs["stream"] = self._step_stream.cuda_stream
                kwargs.setdefault("traj_slots", -64)      # 64 preferred; the library halves it while the ring would exceed 2 GB
        self.local = make(env_id, self.local_num_envs, env_index_base=self.lo, **kwargs)
        # the sentence levels' missions travel as two u64 per env inside the record and become strings again on every rank
        self._sentence = bool(getattr(self.local, "sentence", False))
        if self._sentence:
            from .sentence import SentenceDecoder
            self._decode_sentences = SentenceDecoder()
        self._missions = np.asarray(getattr(self.local, "_missions", ()))
        self._mission_index = {m: i for i, m in enumerate(self._missions.tolist())}
        self._zero_copy = hasattr(self.local, "torch_outputs") and getattr(self.local, "output", "") == "torch"
        self._image_shape = None
        self._image_dtype = None
        self.collectives = 0               # gathers issued so far: one per step / reset / fused launch (tests count them)
        self.collective_calls = 0          # all_gather_into_tensor calls behind them (a step's gather = one per record part)
        self._equal = self.num_envs % self.world_size == 0
        self._gparts = {}                  # persistent buffers of the per-step gather: part -> (receive buffer, padded send buffer or None)

    # ---- the one collective on the path -------------------------------------------------------------------
    def all_gather(self, x):
        """(local_n, ...) -> (num_envs, ...) on every rank.  Equal shards: one all_gather_into_tensor straight into
        the result; ragged shards are padded to the largest shard and trimmed."""
        import torch
        x = _to_tensor(x).contiguous()
        if self.world_size == 1:
            return x
        self.collectives += 1
        self.collective_calls += 1
        tail = tuple(x.shape[1:])
        if self._equal:
            out = torch.empty((self.num_envs,) + tail, dtype=x.dtype, device=x.device)
            self._dist.all_gather_into_tensor(out, x, group=self.group)
            return out
        pad = torch.zeros((self._max_local,) + tail, dtype=x.dtype, device=x.device)
        pad[: x.shape[0]] = x
        out = torch.empty((self.world_size * self._max_local,) + tail, dtype=x.dtype, device=x.device)
        self._dist.all_gather_into_tensor(out, pad, group=self.group)
        return self._compact(out)

    def _compact(self, t):
        """(world * per_rank, ...) rows as gathered -> the (num_envs, ...) global tensor: the tensor itself for equal shards (a view of the
        gather buffer), the ranks' valid rows concatenated for a ragged batch."""
        import torch
        if self._equal:
            return t
        parts = []
        for r in range(self.world_size):
            lo, hi = shard_range(self.num_envs, r, self.world_size)
            parts.append(t[r * self._max_local: r * self._max_local + (hi - lo)])
        return torch.cat(parts, 0)

    def _local_record(self, obs, rew, term, trunc):
        """This rank's step as one contiguous u8 record.  HIP shard: the kernel already wrote it (zero-copy view of
        trajectory slot 0); any other shard (the CPU stand-in of the test-suite): packed here into the same layout."""
        import torch
        image = obs["image"] if isinstance(obs, dict) else obs
        image = _to_tensor(image)
        self._image_shape, self._image_dtype = tuple(image.shape[1:]), image.dtype
        obs_bytes = int(np.prod(self._image_shape)) * image.element_size()
        if self._zero_copy:
            if getattr(self.local._cfg, "null_stream_sync", 0) != 1 and not getattr(self.local, "_stream_arg", None):   # (own step stream: it IS torch's current stream here)
                self.local.sync()          # private non-blocking stream: nothing orders it with the collective's stream
            return self.local.torch_outputs()["record"], obs_bytes
        n = self.local_num_envs
        lay = record_layout(n, obs_bytes, self._sentence)
        rec = torch.zeros(lay["record_bytes"], dtype=torch.uint8)

        def put(name, t):
            b = _to_tensor(t).contiguous().view(torch.uint8).reshape(-1)
            rec[lay[name]: lay[name] + b.numel()] = b

        def put_scalar(name, t):          # field `name` of the n mg_step_scalars entries
            scalar_field(rec, name, n, lay).copy_(_to_tensor(t).reshape(n))
        put("image", image)
        put_scalar("reward", _to_tensor(rew).to(torch.float64))
        put_scalar("terminated", _to_tensor(np.asarray(term, np.uint8)))
        put_scalar("truncated", _to_tensor(np.asarray(trunc, np.uint8)))
        if isinstance(obs, dict):
            put_scalar("direction", _to_tensor(np.asarray(obs["direction"], np.uint8)))
            if self._sentence:
                from .sentence import encode
                words = np.asarray([encode(str(m)) for m in np.asarray(obs["mission"]).tolist()], np.uint64).reshape(n, 2)
                put("sentence", _to_tensor(words.view(np.uint8).reshape(-1)))
            else:
                mis = obs.get("mission_id")
                if mis is None:
                    mis = np.fromiter((self._mission_index[m] for m in np.asarray(obs["mission"]).tolist()), np.uint16, n)
                put_scalar("mission_id", _to_tensor(np.asarray(mis, np.uint16).view(np.int16)))
        return rec, obs_bytes

    def gather_record(self, rec=None, obs_bytes=None):
        """The exchange of one step: this rank's record, part by part, into persistent FIELD-MAJOR buffers on every rank --
        {"image": (world * per_rank, obs_bytes) u8, "scalars": (world * per_rank, 16) u8 [, "sentence": (world * per_rank, 16) u8]},
        per_rank = the largest shard.  One all_gather_into_tensor per part, sent straight out of the record (a short rank of a ragged
        batch sends from its persistent padded buffer).  The buffers are reused by the next step's gather."""
        import torch
        if rec is None:
            rec = self.local.torch_outputs()["record"]
            obs_bytes = int(np.prod(self.local.image_shape))
        W, n, per = self.world_size, self.local_num_envs, self._max_local
        lay = record_layout(n, obs_bytes, self._sentence)
        parts = [("image", lay["image"], obs_bytes), ("scalars", lay["reward"], SCALAR_STRIDE)]
        if self._sentence:
            parts.append(("sentence", lay["sentence"], 16))
        out = {}
        if W > 1:
            self.collectives += 1
        for name, off, row in parts:
            send = rec[off: off + n * row].reshape(n, row)
            if W == 1:
                out[name] = send
                continue
            bufs = self._gparts.get(name)
            if bufs is None or bufs[0].shape != (W * per, row) or bufs[0].device != rec.device:
                recv = torch.empty((W * per, row), dtype=torch.uint8, device=rec.device)
                pad = torch.zeros((per, row), dtype=torch.uint8, device=rec.device) if n != per else None     # allocated ONCE
                bufs = self._gparts[name] = (recv, pad)
            recv, pad = bufs
            if pad is not None:
                pad[:n].copy_(send)
                send = pad
            self.collective_calls += 1
            self._dist.all_gather_into_tensor(recv.reshape(-1), send.reshape(-1), group=self.group)
            out[name] = recv
        return out

    @staticmethod
    def _scalar_view(area, name):
        """Strided view of field `name` of an (rows, 16) u8 array of mg_step_scalars entries."""
        import torch
        if name == "reward":
            return area.view(torch.float64)[:, 0]
        if name == "mission_id":
            return area.view(torch.int16)[:, 6]
        return area[:, _SCALAR_FIELD[name]]

    def _gather_step(self, obs, rew, term, trunc):
        """The step's gather (gather_record); the global per-field tensors are VIEWS of the gathered buffers (equal shards)."""
        import torch
        rec, obs_bytes = self._local_record(obs, rew, term, trunc)
        g = self.gather_record(rec, obs_bytes)
        rows = self.num_envs if self._equal else g["image"].shape[0]

        def scal(name):
            return self._compact(self._scalar_view(g["scalars"], name))
        image = self._compact(g["image"].view(self._image_dtype).reshape((rows,) + tuple(self._image_shape)))
        rew_g = scal("reward")
        term_g = scal("terminated").bool()
        trunc_g = scal("truncated").bool()
        if isinstance(obs, dict) and self._sentence:
            words = self._compact(g["sentence"].view(torch.int64))
            out = {"image": image, "direction": scal("direction").to(torch.int64)}
            if "mission_id" in obs:          # device outputs: the words stay a tensor (minigrid_amd.sentence.decode turns a row into text)
                out["sentence"] = words
            else:
                out["mission"] = self._decode_sentences(words.cpu().numpy().view(np.uint64))
            return out, rew_g, term_g, trunc_g
        if isinstance(obs, dict):
            ids = scal("mission_id")
            out = {"image": image, "direction": scal("direction").to(torch.int64)}
            if "mission_id" in obs:
                out["mission_id"] = ids
            else:
                out["mission"] = self._missions[ids.cpu().numpy()]   # strings do not travel: ids do, mapped back on every rank
            return out, rew_g, term_g, trunc_g
        return image, rew_g, term_g, trunc_g

    def _on_step_stream(self):
        """torch's current stream = the shard's stream, so that torch.distributed orders its collectives against the step kernels.
        The shard's stream is non-blocking, so it is ordered with the CALLER's stream explicitly, both ways: on entry it
        waits for the caller's stream (device action tensors produced there are complete before mg_step reads them), on exit the
        caller's stream waits for it (the tensors handed back -- views of slot 0, the gathered buffer -- are complete for any consumer
        on the caller's stream).  Device-side waits only; the host is not synchronised."""
        import contextlib
        if self._step_stream is None:
            return contextlib.nullcontext()
        import torch

        @contextlib.contextmanager
        def ordered():
            caller = torch.cuda.current_stream(self._step_stream.device)
            self._step_stream.wait_stream(caller)
            try:
                with torch.cuda.stream(self._step_stream):
                    yield
            finally:
                caller.wait_stream(self._step_stream)
        return ordered()

    # ---- the fused rollout: ONE collective per launch, overlapped with the next launch --------------------------------
    def rollout_gather(self, steps: int, action_seed: int = 0, consumer=None) -> int:
        """`steps` lockstep steps of the device policy in fused launches of F = max_fused_steps steps.  After every launch ONE
        all_gather_into_tensor moves the launch's F step records -- a contiguous block of the trajectory ring, zero-copy -- to
        every rank: (world_size, F, record bytes) uint8.  The collective is issued on a dedicated communication stream and ordered
        with events, two ring blocks in rotation, so launch k + 1 runs under the gather of launch k and nothing synchronises the
        host.  `consumer(block, T)` is called per launch with the gathered tensor (on the communication stream when there is one;
        the buffer is reused two launches later; `unpack_block` turns one of its steps into per-field tensors).  Returns the
        number of collectives issued (= launches)."""
        import torch
        loc = self.local
        if int(loc.traj_slots) < 2:
            raise ValueError(f"rollout_gather rotates two blocks of trajectory slots: traj_slots >= 2 is needed, this shard has "
                             f"{int(loc.traj_slots)} (RGB observation modes keep a single slot: use step() with gather=True there)")
        F = max(1, min(int(loc.max_fused_steps), int(loc.traj_slots) // 2))
        W = self.world_size
        # without the own-stream setup (gather=False at construction, a caller-supplied stream, a custom `make`, numpy outputs) nothing
        # orders the shard's launches with the collective's stream or a block's next launch with its previous gather: synchronise
        # the host around every collective instead -- correct, not overlapped
        host_ordered = self._step_stream is None
        rec_bytes = record_layout(self._max_local, int(np.prod(loc.image_shape)), self._sentence)["record_bytes"]
        n_coll, done = 0, 0
        if getattr(self, "_blk", None) is None:
            self._blk = {"gbuf": [None, None], "stepped": [None, None], "gathered": [None, None]}
        st = self._blk
        k = getattr(self, "_blk_next", 0)
        while done < steps:
            T = min(F, steps - done)
            b = k & 1
            slot0 = (b + 1) * F - 1                                   # block b = slots [b F, b F + T) counted from its top
            lo = slot0 - T + 1
            if self._step_stream is not None:
                if st["gathered"][b] is not None:
                    self._step_stream.wait_event(st["gathered"][b])   # the block's previous gather has read it
                with torch.cuda.stream(self._step_stream):
                    loc.rollout_block(T, action_seed, slot0)
                    st["stepped"][b] = self._step_stream.record_event(st["stepped"][b])
                self._comm_stream.wait_event(st["stepped"][b])
            else:
                loc.rollout_block(T, action_seed, slot0)
                if host_ordered and hasattr(loc, "sync"):
                    loc.sync()                                         # the block is complete before the collective reads it
            view = loc.block_view(lo, T)                               # (T, this shard's slot bytes) u8
            with (torch.cuda.stream(self._comm_stream) if self._comm_stream is not None else self._on_step_stream()):
                if W > 1 and view.shape[1] != rec_bytes:                # ragged shard: pad the records to the largest shard's
                    pad = st.get("pad")                                # (one persistent buffer per block, allocated once: the collective of block b reads it
                    if pad is None:                                    #  while block 1 - b is being stepped into)
                        pad = st["pad"] = [torch.zeros((F, rec_bytes), dtype=torch.uint8, device=view.device) for _ in range(2)]
                    pad[b][:T, : view.shape[1]].copy_(view)
                    view = pad[b][:T]
                g = st["gbuf"][b]
                if g is None or g.shape != (W, F, view.shape[1]) or g.device != view.device:
                    g = st["gbuf"][b] = torch.empty((W, F, view.shape[1]), dtype=torch.uint8, device=view.device)
                out = g if T == F else torch.empty((W, T, view.shape[1]), dtype=torch.uint8, device=view.device)
                if W > 1:
                    self.collectives += 1
                    self.collective_calls += 1
                    self._dist.all_gather_into_tensor(out.reshape(-1), view.reshape(-1), group=self.group)
                else:
                    out = view.reshape(1, T, -1)
                n_coll += 1
                if consumer is not None:
                    consumer(out, T)
                if self._comm_stream is not None:
                    st["gathered"][b] = self._comm_stream.record_event(st["gathered"][b])
                elif host_ordered and out.is_cuda:
                    torch.cuda.current_stream(out.device).synchronize()   # ... and has been read before the block is launched into again
            done += T
            k += 1
        self._blk_next = k
        return n_coll

    def finish(self):
        """Wait (host) until every launch and collective issued by rollout_gather has completed."""
        if self._comm_stream is not None:
            self._comm_stream.synchronize()
            self._step_stream.synchronize()

    def unpack_block(self, block, j: int, stacked: bool = False) -> dict:
        """Per-field GLOBAL tensors of step record j of a gathered block ((world, T, bytes) uint8; record j = the launch's step
        T - 1 - j, like trajectory slots): image, reward, terminated, truncated, direction, mission_id, action.
        A block is gathered record-major ([rank][step][image | scalars]: the launch's records ARE the send buffer), so a flat (num_envs, ...)
        tensor of one field is a concatenation over the ranks (a copy).  stacked=True returns zero-copy strided VIEWS of the block instead,
        shaped (world, per_rank, ...): env i of the batch is [i // per_rank, i % per_rank] (equal shards; for a ragged batch the
        rows of a short rank beyond its envs are padding)."""
        import torch
        loc = self.local
        if stacked:
            if not self._equal:
                raise ValueError("unpack_block(stacked=True) needs equal shards (num_envs % world_size == 0): a short rank's records keep their own layout")
            obs_bytes = int(np.prod(loc.image_shape))
            per = self._max_local
            lay = record_layout(per, obs_bytes, self._sentence)       # (the block's records are padded to the largest shard's layout)
            img_dtype = torch.int8 if getattr(loc, "obs_mode", "") == "symbolic" else torch.uint8
            recs = block[:, j]                                          # (world, record bytes), rows F * record bytes apart
            area = recs[:, lay["reward"]: lay["reward"] + per * SCALAR_STRIDE].unflatten(1, (per, SCALAR_STRIDE))
            out = {"image": recs[:, : per * obs_bytes].view(img_dtype).unflatten(1, (per,) + tuple(loc.image_shape))}
            for name in _SCALAR_FIELD:
                if name == "reward":
                    out[name] = area.view(torch.float64)[..., 0]
                elif name == "mission_id":
                    out[name] = area.view(torch.int16)[..., 6]
                else:
                    out[name] = area[..., _SCALAR_FIELD[name]]
            return out
        obs_bytes = int(np.prod(loc.image_shape))
        img_dtype = torch.int8 if getattr(loc, "obs_mode", "") == "symbolic" else torch.uint8
        fields = {"image": (img_dtype, tuple(loc.image_shape)), "reward": (torch.float64, ()), "terminated": (torch.uint8, ()),
                  "truncated": (torch.uint8, ()), "direction": (torch.uint8, ()), "mission_id": (torch.int16, ()), "action": (torch.uint8, ())}
        out = {}
        for name, (dt, tail) in fields.items():
            esz = torch.empty(0, dtype=dt).element_size() * (int(np.prod(tail)) if tail else 1)
            parts = []
            for r in range(self.world_size):
                lo, hi = shard_range(self.num_envs, r, self.world_size)
                n = hi - lo
                lay = record_layout(n, obs_bytes, self._sentence)
                if name in _SCALAR_FIELD:
                    parts.append(scalar_field(block[r, j], name, n, lay))
                else:
                    raw = block[r, j, lay[name]: lay[name] + n * esz]
                    parts.append(raw.view(dt).reshape((n,) + tail))
            out[name] = parts[0] if self.world_size == 1 else torch.cat(parts, 0)
        return out

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
        with self._on_step_stream():
            obs, info = self.local.reset(seed=seed, options=options)     # int seed: the shard adds its env_index_base
            if not self.gather:
                return obs, info
            n = self.local_num_envs
            z = np.zeros(n, np.uint8)
            return self._gather_step(obs, np.zeros(n, np.float64), z, z)[0], info

    def step(self, actions):
        a = self._local_slice(actions, "actions")
        with self._on_step_stream():
            obs, rew, term, trunc, info = self.local.step(a)
            if not self.gather:
                return obs, rew, term, trunc, info
            obs, rew, term, trunc = self._gather_step(obs, rew, term, trunc)
            return obs, rew, term, trunc, info

    def close(self):
        self.local.close()

    def __getattr__(self, name):           # spaces, max_steps, counters(), rollout(), ... come from the local shard
        return getattr(self.local, name)
