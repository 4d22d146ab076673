"""Vector counterparts of the two reference wrappers on the hot path (same names, same semantics).

  ImgObsWrapper   (minigrid/wrappers.py:187-214): observation = obs["image"]; space = the image Box.
  FullyObsWrapper (minigrid/wrappers.py:383-426): obs["image"] = grid.encode() with the agent cell set to
                   (OBJECT_TO_IDX["agent"], COLOR_TO_IDX["red"], agent_dir); other keys unchanged.

The encodes run inside the HIP step kernel (no extra pass): wrapping re-creates the underlying MiniGridVecEnv with
the matching `obs_mode` / `image_only` option, exactly like composing the reference wrappers changes what
`step()` returns.
"""
from __future__ import annotations

from .vector_env import MiniGridVecEnv


def _rebuild(env: MiniGridVecEnv, **changes) -> MiniGridVecEnv:
    kw = dict(obs_mode=env.obs_mode, autoreset_mode=env.metadata["autoreset_mode"],
              rng="philox" if env._cfg.rng_mode == 1 else "pcg64", env_index_base=env.env_index_base,
              max_steps=env.max_steps, output=env.output, image_only=env.image_only)
    kw.update(changes)
    new = MiniGridVecEnv(env.env_id, env.num_envs, **kw)
    env.close()
    return new


def ImgObsWrapper(env: MiniGridVecEnv) -> MiniGridVecEnv:
    """Use the image as the only observation output, no language/mission (wrappers.py:187-214)."""
    return _rebuild(env, image_only=True)


def FullyObsWrapper(env: MiniGridVecEnv) -> MiniGridVecEnv:
    """Fully observable gridworld using a compact grid encoding instead of the agent view (wrappers.py:383-426)."""
    return _rebuild(env, obs_mode="full")
