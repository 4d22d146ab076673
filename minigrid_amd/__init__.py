"""minigrid_amd — MI355X-native lockstep-batched MiniGrid hot path (step/reset/gen_obs + ImgObs/FullyObs encodes).

    import minigrid_amd as mg
    envs = mg.make_vec("MiniGrid-Empty-8x8-v0", num_envs=65536)
    obs, info = envs.reset(seed=0)
    obs, reward, terminated, truncated, info = envs.step(actions)

HIP/gfx950 only: importing works anywhere, creating an env requires an MI355X.
"""
from .registry import EnvSpec, registry, spec  # noqa: F401
from .vector_env import MiniGridVecEnv, make_vec  # noqa: F401
from .wrappers import (DictObservationSpaceWrapper, FullyObsWrapper, ImgObsWrapper, NoDeath,  # noqa: F401
                       OneHotPartialObsWrapper, RGBImgObsWrapper, RGBImgPartialObsWrapper, SymbolicObsWrapper,
                       ViewSizeWrapper)

__version__ = "0.1.0"
