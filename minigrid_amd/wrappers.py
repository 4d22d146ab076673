"""Vector counterparts of the two reference wrappers on the hot path (same names, same semantics).

  ImgObsWrapper   (minigrid/wrappers.py:187-214): observation = obs["image"]; space = the image Box.
  FullyObsWrapper (minigrid/wrappers.py:383-426): obs["image"] = grid.encode() with the agent cell set to
                   (OBJECT_TO_IDX["agent"], COLOR_TO_IDX["red"], agent_dir); other keys unchanged.

The encodes run inside the HIP step kernel (no extra pass): wrapping switches the observation configuration of the
SAME env (obs_mode / view size / ... : `mg_set_obs_config`; `image_only` and the mission vocabulary are host-side),
exactly like composing the reference wrappers changes what `step()` of the one wrapped env returns.
"""
from __future__ import annotations

from .vector_env import MiniGridVecEnv


def _rebuild(env: MiniGridVecEnv, **changes) -> MiniGridVecEnv:
    """The reference's wrappers wrap the SAME env object (wrappers.py:187-214): whatever state it is in -- mid-episode, its np_random
    position -- is what the wrapped env continues from.  Here an observation wrapper is a different encode inside the step kernel:
    the handle's observation configuration is switched in place (`mg_set_obs_config`); the live state -- grids, agent records,
    step counts, missions, the sentence levels' instruction trees, keys hidden in boxes -- and every env's generator position are
    not touched, nothing is allocated besides the new output buffers, and the returned object IS `env`."""
    if changes.get("dict_mission") and not getattr(env, "sentence", False):
        from .mission_vocab import string_to_indices
        for m in env.spec_row.missions:                 # DictObservationSpaceWrapper raises for words outside its vocabulary
            string_to_indices(m)
    return env._reconfigure(**changes)


def ImgObsWrapper(env: MiniGridVecEnv) -> MiniGridVecEnv:
    """Use the image as the only observation output, no language/mission (wrappers.py:187-214)."""
    return _rebuild(env, image_only=True)


def RGBImgObsWrapper(env: MiniGridVecEnv, tile_size: int = 8) -> MiniGridVecEnv:
    """Fully observable RGB image as observation (wrappers.py:287-331): get_frame(highlight=env.highlight, tile_size),
    (height*tile_size, width*tile_size, 3) uint8, blitted on the device from the pre-rendered tile atlas."""
    return _rebuild(env, obs_mode="rgb", tile_size=tile_size)


def RGBImgPartialObsWrapper(env: MiniGridVecEnv, tile_size: int = 8) -> MiniGridVecEnv:
    """Partially observable RGB image as observation (wrappers.py:334-381): get_frame(tile_size, agent_pov=True),
    (view*tile_size, view*tile_size, 3) uint8."""
    return _rebuild(env, obs_mode="rgb_partial", tile_size=tile_size)


def FullyObsWrapper(env: MiniGridVecEnv) -> MiniGridVecEnv:
    """Fully observable gridworld using a compact grid encoding instead of the agent view (wrappers.py:383-426)."""
    return _rebuild(env, obs_mode="full")


def ViewSizeWrapper(env: MiniGridVecEnv, agent_view_size: int = 7) -> MiniGridVecEnv:
    """Customize the agent field of view size (wrappers.py:629-673); like the reference it cannot be combined with
    the fully observable wrappers."""
    if env.obs_mode in ("full", "symbolic"):
        raise ValueError("ViewSizeWrapper cannot be used with fully observable wrappers")
    return _rebuild(env, agent_view_size=agent_view_size)


def OneHotPartialObsWrapper(env: MiniGridVecEnv, tile_size: int = 8) -> MiniGridVecEnv:
    """One-hot encoding of the partially observable agent view: (V, V, 11+6+3) (wrappers.py:217-284)."""
    if env.obs_mode != "partial":
        raise ValueError("OneHotPartialObsWrapper encodes the partial view")
    return _rebuild(env, obs_mode="onehot")


def SymbolicObsWrapper(env: MiniGridVecEnv) -> MiniGridVecEnv:
    """Fully observable grid with a symbolic (x, y, object index or -1) representation (wrappers.py:729-782)."""
    return _rebuild(env, obs_mode="symbolic", agent_view_size=7)


def NoDeath(env: MiniGridVecEnv, no_death_types, death_cost: float = -1.0) -> MiniGridVecEnv:
    """Prevent death in specific cells, paying `death_cost` instead (wrappers.py:809-882)."""
    return _rebuild(env, no_death_types=tuple(no_death_types), death_cost=death_cost)


def DictObservationSpaceWrapper(env: MiniGridVecEnv) -> MiniGridVecEnv:
    """Replace the mission string by its word indices in the Minigrid vocabulary, padded to 50 (wrappers.py:429-554)."""
    return _rebuild(env, dict_mission=True)
