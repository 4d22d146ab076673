"""MiniGridVecEnv — N lockstep MiniGrid environments on one MI355X behind the Gymnasium VectorEnv surface.

Mirrors, for a batch, what the reference exposes per env:
  * `MiniGridEnv.reset(*, seed, options)` / `step(action)` (minigrid/minigrid_env.py:119-157, 525-595),
  * observation dict {image (7,7,3) u8, direction, mission} (minigrid_env.py:72-84, 634-650),
  * `ImgObsWrapper` (wrappers.py:187-214) and `FullyObsWrapper` (wrappers.py:383-426) semantics
    (see minigrid_amd/wrappers.py),
  * errors: unknown action -> ValueError (minigrid_env.py:584-585).
Batch semantics follow gymnasium.vector.VectorEnv (>= 1.0): `reset(seed=int)` seeds env i with seed+i, NEXT_STEP
autoreset by default, rewards float64, terminations/truncations bool, `infos == {}`.

All compute happens in libminigrid_hip.so (HIP, gfx950).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import replace
from typing import Any, Optional, Sequence

import numpy as np

from . import _binding as B
from . import spaces
from .mission_vocab import string_to_indices
from .registry import ENV_DYNOBS, ENV_LEVELGEN, ENV_OPENTWODOORS, EnvSpec, spec as _spec

try:  # subclass the real thing when it exists so isinstance checks pass
    from gymnasium.vector import VectorEnv as _VectorEnvBase  # type: ignore
except Exception:
    class _VectorEnvBase:  # type: ignore
        pass

_AUTORESET = {"next_step": B.AUTORESET_NEXT_STEP, "disabled": B.AUTORESET_DISABLED, "same_step": B.AUTORESET_SAME_STEP}
_OBS_MODES = {"partial": B.OBS_PARTIAL, "full": B.OBS_FULL, "onehot": B.OBS_ONEHOT, "symbolic": B.OBS_SYMBOLIC,
              "rgb_partial": B.OBS_RGB_PARTIAL, "rgb": B.OBS_RGB}
# minigrid/core/constants.py:25-37
OBJECT_TO_IDX = {"unseen": 0, "empty": 1, "wall": 2, "floor": 3, "door": 4, "key": 5, "ball": 6, "box": 7, "goal": 8,
                 "lava": 9, "agent": 10}
_RNG = {"pcg64": B.RNG_PCG64, "philox": B.RNG_PHILOX}


class _DeviceArray:
    """Zero-copy view of a library-owned device buffer (`__cuda_array_interface__` v3; torch.as_tensor accepts it)."""

    def __init__(self, ptr: int, shape, typestr: str, owner, strides=None):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 3, "strides": None if strides is None else tuple(strides)}
        self._owner = owner  # keep the env alive


class MiniGridVecEnv(_VectorEnvBase):
    metadata = {"render_modes": [], "autoreset_mode": "next_step"}

    def __init__(self, env_id: str, num_envs: int, *, obs_mode: str = "partial", device: Optional[int] = None,
                 autoreset_mode: str = "next_step", rng: str = "pcg64", env_index_base: int = 0,
                 max_steps: Optional[int] = None, stream: Optional[int] = None, output: str = "numpy",
                 image_only: bool = False, agent_view_size: int = 7, no_death_types: Sequence[str] = (),
                 death_cost: float = -1.0, dict_mission: bool = False, tile_size: int = 8, highlight: bool = True,
                 spare_ring: int = 0, traj_slots: int = 0, stuck_place_agent: str = "raise", final_obs: bool = False,
                 babyai_done_actions=None):
        if obs_mode not in _OBS_MODES:
            raise ValueError(f"obs_mode must be one of {sorted(_OBS_MODES)}")
        # ViewSizeWrapper.__init__ asserts (wrappers.py:650-651)
        assert agent_view_size % 2 == 1
        assert agent_view_size >= 3
        if agent_view_size > 15:
            raise ValueError("agent_view_size up to 15 is supported on the accelerated path")
        assert "goal" not in no_death_types, "goal cannot be a death cell"      # NoDeath.__init__ (wrappers.py:854)
        if output not in ("numpy", "torch"):
            raise ValueError("output must be 'numpy' or 'torch'")
        # RoomGrid.place_agent loops without a bound (roomgrid.py:327-332): when every free cell of the agent's room faces an object the
        # reference never returns from reset() (BabyAI-SynthS5R2-v0: about 0.4 % of the episodes).  "raise": RecursionError when an env
        # reaches such an episode; "redraw" (LevelGen levels only): the attempt ends like the RecursionError the level's retry loop catches
        # and the redrawn map is accepted -- usable at batch sizes where some env always meets the case, but not a reference behaviour.
        if stuck_place_agent not in ("raise", "redraw"):
            raise ValueError("stuck_place_agent must be 'raise' or 'redraw'")
        # autoreset_mode="same_step" resets inside the step kernel: the returned observation is the new episode's first one and the
        # terminal observation is gone (what Gymnasium 0.28/0.29 users saw as info["final_observation"]).  final_obs=True keeps it, the
        # way Gymnasium 1.x's SAME_STEP vector envs report it -- info["final_obs"] (object array of observation dicts) and
        # info["_final_obs"] (mask) -- by composition instead: the step kernel runs with NEXT_STEP semantics (the terminal observation
        # is its output), and the envs that finished are reset and observed by a second, masked launch before step() returns.
        # One launch per step(): the fused entry points refuse this mode.
        if final_obs and autoreset_mode != "same_step":
            raise ValueError("final_obs=True belongs to autoreset_mode='same_step'")
        self._final_obs = bool(final_obs)
        # The in-kernel SAME_STEP of the sentence levels (their episodes end in the verifier) and of DynamicObstacles (its reset draws on the stream its
        # steps consume) is built for the default 7x7x3 observation (mg_api.hip validate_obs_cfg refuses the other modes).  For those combinations
        # same_step is composed here exactly like final_obs -- a NEXT_STEP launch, then a masked reset of the envs that finished, each continuing its
        # own stream: the same observations, rewards and stream positions (tests/test_gpu_roll.py), two launches per step(), no fused entry points.
        _kind = _spec(env_id).env_kind
        self._composed_same_step = (autoreset_mode == "same_step" and not final_obs and
                                    (_kind == ENV_DYNOBS or ENV_OPENTWODOORS <= _kind <= ENV_LEVELGEN) and
                                    not (obs_mode == "partial" and int(agent_view_size) == 7))
        # envs/babyai/core/verifier.py:26: the reference reads BABYAI_DONE_ACTIONS when it is imported ("any non-empty value"); the same
        # variable is the default here, the argument overrides it
        if babyai_done_actions is None:
            import os
            babyai_done_actions = bool(os.environ.get("BABYAI_DONE_ACTIONS", False))
        # True: the reference stepped with integer actions (env.step(6), SyncVectorEnv's numpy elements); "enum": stepped with Actions members
        # (env.step(env.actions.done)), which alone take AndInstr.verify's `action is self.env.actions.done` branch (verifier.py:561-563)
        self.babyai_done_actions = "enum" if babyai_done_actions == "enum" else bool(babyai_done_actions)
        # what pickling needs to build the same env again (__getstate__)
        self._ctor = dict(env_id=env_id, num_envs=int(num_envs), autoreset_mode=autoreset_mode, rng=rng, env_index_base=int(env_index_base),
                          max_steps=max_steps, output=output, spare_ring=int(spare_ring), traj_slots=int(traj_slots),
                          stuck_place_agent=stuck_place_agent, final_obs=bool(final_obs), babyai_done_actions=self.babyai_done_actions)
        s: EnvSpec = _spec(env_id)
        if stuck_place_agent == "redraw":
            if s.env_kind != ENV_LEVELGEN:
                raise ValueError("stuck_place_agent='redraw' applies to the LevelGen levels (PickupLoc, GoToSeq*, Synth*, *BossLevel*)")
            s = replace(s, num_crossings=s.num_crossings | 1 << 10)
        if max_steps is not None:
            if not isinstance(max_steps, int):
                raise AssertionError(f"The argument max_steps must be an integer, got: {type(max_steps)}")  # minigrid_env.py:102-104
            s = s.with_max_steps(max_steps)
        self.spec_row = s
        self.env_id = env_id
        self.num_envs = int(num_envs)
        self.obs_mode = obs_mode
        self.output = output
        self.image_only = bool(image_only)
        self.env_index_base = int(env_index_base)
        self.metadata = dict(type(self).metadata, autoreset_mode=autoreset_mode)
        self._lib = B.load()
        if self._lib.mg_device_count() < 1:
            raise B.MiniGridHipError("no HIP device visible: minigrid_amd has no CPU fallback (MI355X/gfx950 required)")
        self.agent_view_size = int(agent_view_size)
        self.no_death_types = tuple(no_death_types)
        self.death_cost = float(death_cost)
        self.dict_mission = bool(dict_mission)
        no_death_mask = 0
        for t in self.no_death_types:
            no_death_mask |= 1 << OBJECT_TO_IDX[t]
        cfg = B.MgConfig(
            abi_version=B.MG_ABI_VERSION, env_kind=s.env_kind, width=s.width, height=s.height, max_steps=s.max_steps,
            see_through_walls=int(s.see_through_walls), agent_view_size=self.agent_view_size,
            no_death_mask=no_death_mask, death_cost=self.death_cost,
            obs_mode=_OBS_MODES[obs_mode], autoreset_mode=_AUTORESET["next_step" if (final_obs or self._composed_same_step) else autoreset_mode],
            rng_mode=_RNG[rng], num_envs=self.num_envs, agent_start_x=s.agent_start[0], agent_start_y=s.agent_start[1],
            agent_start_dir=s.agent_start[2], num_crossings=s.num_crossings, obstacle_type=s.obstacle_type,
            num_dists=s.num_dists, strip2_row=s.strip2_row, room_size=s.room_size, random_length=int(s.random_length),
            env_index_base=self.env_index_base, tile_size=int(tile_size), rgb_highlight=int(bool(highlight)),
            spare_ring=int(spare_ring), traj_slots=int(traj_slots), babyai_done_actions=2 if self.babyai_done_actions == "enum" else int(self.babyai_done_actions))
        self.tile_size, self.highlight = int(tile_size), bool(highlight)
        self.spare_ring, self.traj_slots_arg = int(spare_ring), int(traj_slots)
        self.rng_kind = rng
        if output == "torch" and stream is None:
            # outputs are handed out as torch tensors: run stream-ordered with torch.  A non-default current stream is
            # borrowed; the legacy NULL stream (torch's default) cannot be passed as a handle, so the library's own
            # stream is created blocking, which orders it with NULL-stream work in both directions.
            import torch
            ts = int(torch.cuda.current_stream().cuda_stream)
            if ts:
                stream = ts
            else:
                cfg.null_stream_sync = 1
        self._cfg = cfg
        self._stream_arg = stream
        h = C.c_void_p()
        rc = self._lib.mg_create(C.byref(cfg), -1 if device is None else int(device), stream, C.byref(h))
        B.check(rc, None)
        self._h = h
        if device is None:
            try:
                import torch
                device = torch.cuda.current_device()
            except Exception:
                device = 0
        self.device = int(device)
        self.width, self.height, self.max_steps = s.width, s.height, s.max_steps
        self._missions = np.asarray(s.missions)
        self._seeded = False
        self._bind_outputs()

    def _bind_outputs(self):
        """Everything that follows from the observation configuration of the handle: output pointers, shapes, spaces, host staging.
        Called at construction and again after `_reconfigure` (an observation wrapper applied to this env)."""
        s = self.spec_row
        obs_mode, image_only = self.obs_mode, self.image_only
        outs = B.MgOutputs()
        B.check(self._lib.mg_get_outputs(self._h, C.byref(outs)), self._h)
        self._outs = outs
        self.traj_slots, self.max_fused_steps = int(outs.traj_slots), int(outs.max_fused_steps)
        v = self.agent_view_size
        self.image_shape = {"partial": (v, v, 3), "full": (s.width, s.height, 3), "onehot": (v, v, 20),
                            "symbolic": (s.width, s.height, 3),
                            # RGBImgPartialObsWrapper / RGBImgObsWrapper spaces (wrappers.py:357-368, 307-323): rows x columns x 3
                            "rgb_partial": (v * self.tile_size, v * self.tile_size, 3),
                            "rgb": (s.height * self.tile_size, s.width * self.tile_size, 3)}[obs_mode]
        # sentence levels (instruction trees: GoToSeq, Synth*, BossLevel*, OpenTwoDoors, ...): the mission arrives as data
        self.sentence = bool(outs.sentence)
        if self.sentence:
            from .sentence import SentenceDecoder
            self._decode_sentences = SentenceDecoder()
            self._h_sent = np.empty((self.num_envs, 2), np.uint64)
        # DictObservationSpaceWrapper (wrappers.py:429-554): mission string -> padded word-index vector, per mission id.  Its fixed
        # vocabulary lacks some BabyAI words ("next", "on", "left", ...): like the reference, wrapping such a level raises ValueError
        self._mission_tokens = (np.asarray([string_to_indices(m) for m in s.missions], np.int64)
                                if self.dict_mission and not self.sentence else None)
        # spaces (minigrid_env.py:63, 72-84; FullyObsWrapper wrappers.py:404-417; ImgObsWrapper :211)
        # image spaces as the reference wrappers declare them (wrappers.py:263-267, 404-411, 655-662, 749-758)
        img_high = 10 if obs_mode == "symbolic" else 255
        img = spaces.Box(0, img_high, self.image_shape, np.uint8)
        self.single_action_space = spaces.Discrete(7)
        if image_only:
            self.single_observation_space = img
            self.observation_space = spaces.Box(0, 255, (self.num_envs,) + self.image_shape, np.uint8)
        else:
            from .mission_vocab import MAX_WORDS_IN_MISSION, minigrid_words
            mission_space = (spaces.MultiDiscrete([len(minigrid_words())] * MAX_WORDS_IN_MISSION) if self.dict_mission
                             else spaces.MissionSpace(s.missions))       # wrappers.py:464-472 / minigrid_env.py:72-84
            self.single_observation_space = spaces.Dict({"image": img, "direction": spaces.Discrete(4),
                                                         "mission": mission_space})
            self.observation_space = spaces.Dict({
                "image": spaces.Box(0, 255, (self.num_envs,) + self.image_shape, np.uint8),
                "direction": spaces.MultiDiscrete([4] * self.num_envs),
                "mission": spaces.MissionSpace(s.missions)})
        self.action_space = spaces.MultiDiscrete([7] * self.num_envs)
        # host staging (numpy output mode)
        n = self.num_envs
        self._h_obs = np.empty((n,) + self.image_shape, np.int8 if obs_mode == "symbolic" else np.uint8)
        self._h_rew = np.empty(n, np.float64)
        self._h_term = np.empty(n, np.uint8)
        self._h_trunc = np.empty(n, np.uint8)
        self._h_dir = np.empty(n, np.uint8)
        self._h_mis = np.empty(n, np.uint16)
        self._torch_views = None

    # ------------------------------------------------------------------ helpers
    @property
    def handle(self):
        return self._h

    @property
    def unwrapped(self):
        return self

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def device_outputs(self, slot: int = 0) -> dict:
        """Zero-copy device views of the output buffers (valid until close(); rewritten by every step/reset).
        `slot` k of the trajectory ring = the step k calls before the last one (rollout(fused=True) / step_many)."""
        if not 0 <= slot < self.traj_slots:
            raise ValueError(f"slot must be in 0..{self.traj_slots - 1}")
        n, o, off = self.num_envs, self._outs, slot * int(self._outs.slot_bytes)
        st = (int(o.scalar_stride),)        # the scalars of a step are one 16-byte mg_step_scalars per env (ABI 3): strided views of its fields
        return {"image": _DeviceArray(o.obs + off, (n,) + self.image_shape, "|i1" if self.obs_mode == "symbolic" else "|u1", self),
                "reward": _DeviceArray(o.reward + off, (n,), "<f8", self, st),
                "terminated": _DeviceArray(o.terminated + off, (n,), "|u1", self, st),
                "truncated": _DeviceArray(o.truncated + off, (n,), "|u1", self, st),
                "direction": _DeviceArray(o.direction + off, (n,), "|u1", self, st),
                "mission_id": _DeviceArray(o.mission_id + off, (n,), "<i2", self, st),     # 14-bit ids; int16 is the portable 2-byte dtype
                "action": _DeviceArray(o.action + off, (n,), "|u1", self, st),
                # the whole step as ONE contiguous byte record (obs | (N) x mg_step_scalars {reward, terminated, truncated, direction,
                # action, mission id}): what a multi-GPU consumer all-gathers (minigrid_amd/sharded.py)
                "record": _DeviceArray(o.obs + off, (int(o.record_bytes),), "|u1", self),
                # sentence levels: the mission as data (minigrid_amd/sentence.py decodes it), two u64 per env
                **({"sentence": _DeviceArray(o.sentence + off, (n, 2), "<u8", self)} if o.sentence else {})}

    def record_layout(self) -> dict:
        """Byte offsets of env 0's fields inside a step record (device_outputs()["record"]); the scalar fields of env i lie
        i * scalar_stride bytes further (one 16-byte mg_step_scalars per env)."""
        o = self._outs
        return {"image": 0, "reward": o.reward - o.obs, "terminated": o.terminated - o.obs, "truncated": o.truncated - o.obs,
                "direction": o.direction - o.obs, "mission_id": o.mission_id - o.obs, "action": o.action - o.obs,
                "scalar_stride": int(o.scalar_stride),
                "record_bytes": int(o.record_bytes), **({"sentence": o.sentence - o.obs} if o.sentence else {})}

    def torch_outputs(self, slot: int = 0) -> dict:
        """The same buffers as torch CUDA tensors (no copy)."""
        if self._torch_views is None:
            self._torch_views = {}
        if slot not in self._torch_views:
            import torch
            dev = torch.device("cuda", self.device)
            self._torch_views[slot] = {k: torch.as_tensor(v, device=dev) for k, v in self.device_outputs(slot).items()}
        return self._torch_views[slot]

    def trajectory(self, slot: int, image: bool = True):
        """Host copy of trajectory slot `slot`: (image, reward, terminated, truncated, direction, mission_id, action).
        image=False skips the (large) observation copy and returns None in its place."""
        n = self.num_envs
        img = np.empty((n,) + self.image_shape, np.int8 if self.obs_mode == "symbolic" else np.uint8) if image else None
        rew = np.empty(n, np.float64)
        u8 = [np.empty(n, np.uint16 if k == 3 else np.uint8) for k in range(5)]      # mission ids are 16 bits wide
        rc = self._lib.mg_copy_slot(self._h, int(slot), None if img is None else self._p(img), self._p(rew), *[self._p(a) for a in u8])
        B.check(rc, self._h)
        return img, rew, u8[0].astype(bool), u8[1].astype(bool), u8[2], u8[3], u8[4]

    def trajectory_missions(self, slot: int):
        """The mission strings of trajectory slot `slot` (sentence levels: decoded from the slot's mission words; else by mission id)."""
        if self.sentence:
            buf = np.empty((self.num_envs, 2), np.uint64)
            B.check(self._lib.mg_copy_sentence(self._h, int(slot), self._p(buf)), self._h)
            return self._decode_sentences(buf)
        return self._missions[self.trajectory(slot, image=False)[5]]

    def sync(self):
        B.check(self._lib.mg_sync(self._h), self._h)

    def _collect(self):
        if self.output == "torch":
            t = self.torch_outputs()
            image = t["image"]
            if self.image_only:
                obs = image
            else:
                obs = {"image": image, "direction": t["direction"], "mission_id": t["mission_id"]}
            return obs, t["reward"], t["terminated"].bool(), t["truncated"].bool()
        rc = self._lib.mg_copy_outputs(self._h, self._p(self._h_obs), self._p(self._h_rew), self._p(self._h_term),
                                       self._p(self._h_trunc), self._p(self._h_dir), self._p(self._h_mis))
        B.check(rc, self._h)
        # SymbolicObsWrapper returns numpy's default integer array (np.mgrid, wrappers.py:773-780); the device emits int8
        image = self._h_obs.astype(np.int64) if self.obs_mode == "symbolic" else self._h_obs.copy()
        if self.image_only:
            obs = image
        else:
            if self.sentence:
                B.check(self._lib.mg_copy_sentence(self._h, 0, self._p(self._h_sent)), self._h)
                mission = self._decode_sentences(self._h_sent)
                if self.dict_mission:
                    mission = np.asarray([string_to_indices(m) for m in mission], np.int64)
            else:
                mission = self._mission_tokens[self._h_mis] if self.dict_mission else self._missions[self._h_mis]
            obs = {"image": image, "direction": self._h_dir.astype(np.int64), "mission": mission}
        return obs, self._h_rew.copy(), self._h_term.astype(bool), self._h_trunc.astype(bool)

    # ------------------------------------------------------------------ Gymnasium VectorEnv surface
    def reset(self, *, seed: Any = None, options: Optional[dict] = None):
        n = self.num_envs
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = np.ascontiguousarray(options["reset_mask"], dtype=np.uint8)
            if mask.shape != (n,):
                raise ValueError("options['reset_mask'] must have shape (num_envs,)")
        seeds = None
        if seed is None:
            if not self._seeded:      # like a never-seeded gymnasium env: fresh OS entropy per env
                seeds = np.random.SeedSequence().generate_state(n, np.uint64)
        elif isinstance(seed, (int, np.integer)):
            if seed < 0:
                raise ValueError(f"Seed must be a non-negative integer, got {seed}")
            seeds = (np.uint64(seed) + np.uint64(self.env_index_base) + np.arange(n, dtype=np.uint64))
        else:
            seq = list(seed)
            if len(seq) != n:
                raise ValueError("seed list must have one entry per env")
            if any(s is None for s in seq):
                m2 = np.array([s is not None for s in seq], np.uint8)
                if mask is not None:
                    m2 &= mask
                # envs with seed None continue their stream; reseed the others in a second call
                none_mask = np.array([s is None for s in seq], np.uint8) if mask is None else (np.array([s is None for s in seq], np.uint8) & mask)
                if none_mask.any():
                    B.check(self._lib.mg_reset(self._h, None, self._p(none_mask)), self._h)
                mask = m2
                seq = [0 if s is None else s for s in seq]
            seeds = np.asarray(seq, dtype=np.uint64)
        self._seeded = True
        rc = self._lib.mg_reset(self._h, None if seeds is None else self._p(np.ascontiguousarray(seeds)),
                                None if mask is None else self._p(mask))
        B.check(rc, self._h)
        obs, _, _, _ = self._collect()
        return obs, {}

    def step(self, actions):
        if self.output == "torch" and hasattr(actions, "data_ptr"):
            a = actions
            if a.dtype not in _TORCH_ACT or a.numel() != self.num_envs or not a.is_contiguous():
                raise ValueError("actions tensor must be contiguous uint8/int32/int64 of length num_envs")
            rc = self._lib.mg_step(self._h, C.c_void_p(a.data_ptr()), _TORCH_ACT[a.dtype], 1 if a.is_cuda else 0)
        else:
            a = np.asarray(actions)
            if a.shape != (self.num_envs,):
                raise ValueError(f"actions must have shape ({self.num_envs},), got {a.shape}")
            if a.dtype == np.uint8:
                dt = B.ACT_U8
            elif a.dtype == np.int32:
                dt = B.ACT_I32
            else:
                a = a.astype(np.int64, copy=False)
                dt = B.ACT_I64
            a = np.ascontiguousarray(a)
            self._last_actions = a          # the async H2D copy may still read it after this call returns
            rc = self._lib.mg_step(self._h, self._p(a), dt, 0)
        B.check(rc, self._h)
        obs, rew, term, trunc = self._collect()
        if self._final_obs:
            return self._same_step_with_final_obs(obs, rew, term, trunc)
        if self._composed_same_step:
            return self._same_step_with_final_obs(obs, rew, term, trunc, want_final=False)
        return obs, rew, term, trunc, {}

    def _same_step_with_final_obs(self, obs, rew, term, trunc, want_final=True):
        """SAME_STEP by composition (see __init__): `obs` is the step kernel's NEXT_STEP output, i.e. the terminal observation for the envs
        that just finished.  Those envs take their next episode now (masked reset, each continuing its own stream) and the returned
        observation shows it; the terminal one goes to info["final_obs"]."""
        n = self.num_envs
        if self.output == "torch":
            import torch
            done = term | trunc
            if not bool(done.any()):
                return obs, rew, term, trunc, {}
            # the outputs are views of trajectory slot 0, which the reset launch rewrites: keep what this step reported
            rew, term, trunc = rew.clone(), term.clone(), trunc.clone()
            if want_final:
                idx = torch.nonzero(done).flatten()
                final = obs[idx].clone() if self.image_only else {k: v[idx].clone() for k, v in obs.items()}
            mask = done.to(torch.uint8).cpu().numpy()
            B.check(self._lib.mg_reset(self._h, None, self._p(np.ascontiguousarray(mask))), self._h)
            new_obs, _, _, _ = self._collect()
            if not want_final:          # (composed SAME_STEP without final_obs=True: only the masked reset and the new observations -- ADVICE r5)
                return new_obs, rew, term, trunc, {}
            # torch outputs: the terminal observations of the finished envs as COMPACT tensors (row j belongs to env final_obs_indices[j])
            # instead of Gymnasium's object array of per-env dicts; the mask keys are the same as on the numpy path.  (The `.any()`
            # above is a host synchronisation per step: this mode composes two launches on the host and needs to know whether the second
            # one is due; the fused entry points are the path without host round trips.)
            return new_obs, rew, term, trunc, {"final_obs": final, "final_obs_indices": idx, "_final_obs": done,
                                               "final_info": {}, "_final_info": done}
        done = term | trunc
        if not done.any():
            return obs, rew, term, trunc, {}
        if want_final:
            idx = np.flatnonzero(done)
            fo = np.full(n, None, dtype=object)
            for i in idx:
                fo[i] = obs[i] if self.image_only else {"image": obs["image"][i], "direction": obs["direction"][i], "mission": obs["mission"][i]}
        else:
            rew, term, trunc = rew.copy(), term.copy(), trunc.copy()      # (views of trajectory slot 0, which the reset launch rewrites)
        B.check(self._lib.mg_reset(self._h, None, self._p(np.ascontiguousarray(done.astype(np.uint8)))), self._h)
        new_obs, _, _, _ = self._collect()
        if not want_final:
            return new_obs, rew, term, trunc, {}
        return new_obs, rew, term, trunc, {"final_obs": fo, "_final_obs": done, "final_info": np.full(n, None, dtype=object), "_final_info": done}

    def _no_fused_with_final_obs(self):
        if self._composed_same_step:
            raise ValueError("autoreset_mode='same_step' of this level is composed from two launches per step() for this observation mode "
                             "(the in-kernel form serves the default 7x7x3 observation): the fused entry points need that observation, or 'next_step'")
        if self._final_obs:
            raise ValueError("final_obs=True reports terminal observations from step(), one launch at a time; "
                             "the fused entry points need autoreset_mode='same_step' without it (or 'next_step')")

    def rollout(self, steps: int, action_seed: int = 0, fused: bool = False):
        """`steps` lockstep steps under a uniform-random policy generated on the device (the loop of
        minigrid/benchmark.py:36-43).  fused: up to `max_fused_steps` steps per kernel launch, the grids resident in
        LDS; step j writes trajectory slot (steps-1-j) % traj_slots, so slot 0 is the last step."""
        self._no_fused_with_final_obs()
        B.check(self._lib.mg_rollout(self._h, int(steps), int(action_seed), int(fused)), self._h)

    def rollout_block(self, steps: int, action_seed: int = 0, slot0: Optional[int] = None):
        """ONE fused launch of `steps` <= max_fused_steps steps of the device policy; step j lands in trajectory slot slot0 - j
        (default slot0 = steps - 1), so the launch's step records are the contiguous slots [slot0 - steps + 1, slot0]."""
        self._no_fused_with_final_obs()
        B.check(self._lib.mg_rollout_block(self._h, int(steps), int(action_seed), int(steps - 1 if slot0 is None else slot0)), self._h)

    def block_view(self, slot_lo: int, nslots: int):
        """Zero-copy torch view of `nslots` consecutive trajectory slots starting at slot_lo: (nslots, slot_bytes) uint8."""
        import torch
        if not (0 <= slot_lo and slot_lo + nslots <= self.traj_slots):
            raise ValueError("block outside the trajectory ring")
        sb = int(self._outs.slot_bytes)
        arr = _DeviceArray(self._outs.obs + slot_lo * sb, (nslots, sb), "|u1", self)
        return torch.as_tensor(arr, device=torch.device("cuda", self.device))

    def step_many(self, actions):
        """The fused loop for caller-supplied actions: uint8 (T, num_envs), numpy or a CUDA tensor.  Identical to T
        step() calls; outputs of the step k calls before the last are in trajectory slot k (trajectory() /
        torch_outputs(k))."""
        self._no_fused_with_final_obs()
        if hasattr(actions, "data_ptr"):
            import torch
            a = actions
            if a.dtype != torch.uint8 or a.dim() != 2 or a.shape[1] != self.num_envs or not a.is_contiguous():
                raise ValueError("actions must be a contiguous uint8 tensor of shape (T, num_envs)")
            rc = self._lib.mg_step_many(self._h, C.c_void_p(a.data_ptr()), int(a.shape[0]), 1 if a.is_cuda else 0)
            self._last_actions = a
        else:
            a = np.ascontiguousarray(actions, dtype=np.uint8)
            if a.ndim != 2 or a.shape[1] != self.num_envs:
                raise ValueError("actions must have shape (T, num_envs)")
            self._last_actions = a
            rc = self._lib.mg_step_many(self._h, self._p(a), int(a.shape[0]), 0)
            B.check(rc, self._h)
            self.sync()                    # the staged host buffer is reused chunk by chunk
            return
        B.check(rc, self._h)

    def _reconfigure(self, **changes):
        """Apply an observation wrapper to THIS env, the way the reference's wrappers wrap the same env object (wrappers.py:187-214):
        the handle's observation configuration changes in place (mg_set_obs_config) -- grids, agent records, instruction trees,
        hidden box contents and every env's np_random position are untouched, nothing is re-created."""
        py_only = {"image_only", "dict_mission"}
        cfg_map = {"obs_mode": ("obs_mode", lambda v: _OBS_MODES[v]), "agent_view_size": ("agent_view_size", int),
                   "tile_size": ("tile_size", int), "highlight": ("rgb_highlight", lambda v: int(bool(v))),
                   "death_cost": ("death_cost", float)}
        cfg = B.MgConfig.from_buffer_copy(self._cfg)
        for k, v in changes.items():
            if k in py_only:
                continue
            if k == "no_death_types":
                assert "goal" not in v, "goal cannot be a death cell"          # NoDeath.__init__ (wrappers.py:854)
                m = 0
                for t in v:
                    m |= 1 << OBJECT_TO_IDX[t]
                cfg.no_death_mask = m
            elif k in cfg_map:
                name, conv = cfg_map[k]
                setattr(cfg, name, conv(v))
            else:
                raise TypeError(f"cannot change {k!r} on a live env")
        if "agent_view_size" in changes:
            v = int(changes["agent_view_size"])
            assert v % 2 == 1 and v >= 3                                          # ViewSizeWrapper.__init__ (wrappers.py:650-651)
            if v > 15:
                raise ValueError("agent_view_size up to 15 is supported on the accelerated path")
        B.check(self._lib.mg_set_obs_config(self._h, C.byref(cfg)), self._h)
        self._cfg = cfg
        for k, v in changes.items():
            setattr(self, {"no_death_types": "no_death_types"}.get(k, k), tuple(v) if k == "no_death_types" else v)
        self.image_only, self.dict_mission = bool(self.image_only), bool(self.dict_mission)
        self._torch_views = None
        self._bind_outputs()
        return self

    def close(self, **kwargs):
        if getattr(self, "_h", None):
            self._lib.mg_destroy(self._h)
            self._h = None
            self._torch_views = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ state exchange (checkpoint / parity harness)
    def get_state(self):
        grid = np.empty((self.num_envs, self.width, self.height, 3), np.uint8)
        agent = np.empty((self.num_envs, 8), np.int32)
        B.check(self._lib.mg_get_state(self._h, self._p(grid), self._p(agent)), self._h)
        return grid, agent

    def set_state(self, grid, agent):
        grid = np.ascontiguousarray(grid, np.uint8)
        agent = np.ascontiguousarray(agent, np.int32)
        if grid.shape != (self.num_envs, self.width, self.height, 3) or agent.shape != (self.num_envs, 8):
            raise ValueError("bad state shapes")
        B.check(self._lib.mg_set_state(self._h, self._p(grid), self._p(agent)), self._h)

    def save_state(self) -> bytes:
        """Lossless checkpoint of the live batch (mg_save_state): grids incl. what boxes hide, agent records, level words, every env's
        np_random position, the sentence levels' instruction trees -- what pickling a reference env carries."""
        nb = C.c_int64()
        B.check(self._lib.mg_state_size(self._h, C.byref(nb)), self._h)
        buf = np.empty(int(nb.value), np.uint8)
        B.check(self._lib.mg_save_state(self._h, self._p(buf), nb), self._h)
        return buf.tobytes()

    def load_state(self, blob: bytes):
        """Continue from a checkpoint taken from an env of the same level, grid and batch size (any observation configuration)."""
        buf = np.frombuffer(blob, np.uint8).copy()
        B.check(self._lib.mg_load_state(self._h, self._p(buf), C.c_int64(buf.size)), self._h)
        self._seeded = True

    def __getstate__(self):
        # pickling = the constructor arguments as they are now (observation wrappers included) + the checkpoint
        kw = dict(self._ctor, obs_mode=self.obs_mode, image_only=self.image_only, agent_view_size=self.agent_view_size,
                  no_death_types=tuple(self.no_death_types), death_cost=self.death_cost, dict_mission=self.dict_mission,
                  tile_size=self.tile_size, highlight=self.highlight, device=self.device)
        return {"ctor": kw, "seeded": self._seeded, "state": self.save_state()}

    def __setstate__(self, st):
        kw = dict(st["ctor"])
        self.__init__(kw.pop("env_id"), kw.pop("num_envs"), **kw)
        self.load_state(st["state"])
        self._seeded = st["seeded"]

    def get_rng_state(self):
        r = np.empty((self.num_envs, 5), np.uint64)
        B.check(self._lib.mg_get_rng(self._h, self._p(r)), self._h)
        return r

    def set_rng_state(self, r):
        r = np.ascontiguousarray(r, np.uint64)
        B.check(self._lib.mg_set_rng(self._h, self._p(r)), self._h)

    @property
    def spare_ring_depth(self) -> int:
        """The EFFECTIVE spare-episode ring depth of this handle (mg_ring_depth): `spare_ring` when given, else the level's default, halved while the
        ring would exceed min(32 GB, a quarter of the free device memory).  `spare_ring` itself stays the constructor argument (0 = default)."""
        return int(self._lib.mg_ring_depth(self._h))

    def counters(self) -> dict:
        c = np.zeros(4, np.uint64)
        B.check(self._lib.mg_get_counters(self._h, self._p(c)), self._h)
        return {"env_steps": int(c[0]), "episodes": int(c[1]), "maps_generated": int(c[2]), "generator_retries": int(c[3])}

    def timer_start(self):
        B.check(self._lib.mg_timer_start(self._h), self._h)

    def timer_stop(self) -> float:
        ms = C.c_float()
        B.check(self._lib.mg_timer_stop(self._h, C.byref(ms)), self._h)
        return float(ms.value)


try:
    import os as _os
    if _os.environ.get("MINIGRID_AMD_NO_TORCH", "0") == "1":       # a torch-free process (numpy outputs only): see _binding.load
        raise ImportError
    import torch as _torch
    _TORCH_ACT = {_torch.uint8: B.ACT_U8, _torch.int32: B.ACT_I32, _torch.int64: B.ACT_I64}
except Exception:  # pragma: no cover
    _TORCH_ACT = {}


def make_vec(env_id: str, num_envs: int, **kwargs) -> MiniGridVecEnv:
    """`gymnasium.make_vec(id, num_envs)` for the accelerated ids (minigrid/__init__.py registry rows)."""
    return MiniGridVecEnv(env_id, num_envs, **kwargs)
