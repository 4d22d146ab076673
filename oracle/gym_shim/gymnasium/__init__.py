"""Minimal stand-in for the third-party `gymnasium` package (TEST INFRASTRUCTURE ONLY).

gymnasium is not installed in the build container and there is no network.  The
reference (/root/reference, read-only) imports it, so to run the *unmodified*
reference as the parity oracle we provide the small slice of gymnasium's public
API the reference's hot path touches (SURVEY.md §8c lists the call sites):
`Env.reset(seed=)` seeding -> numpy Generator(PCG64(SeedSequence(seed))),
`spaces.{Space,Box,Discrete,Dict,MultiDiscrete}`, `core.{Wrapper,ObservationWrapper,
ActionWrapper}`, `envs.registration.{register,registry,make}`, `utils.seeding.np_random`,
`logger.warn`, `error.DependencyNotInstalled`.

Nothing in the product (`minigrid_amd/`) imports this.  It is written from
gymnasium's documented behaviour, not copied from its sources.
"""
from __future__ import annotations

from . import error, logger, spaces, utils  # noqa: F401
from .core import ActionWrapper, Env, ObservationWrapper, Wrapper  # noqa: F401
from .envs import registration as _registration
from .envs.registration import make, register, registry  # noqa: F401
from . import envs  # noqa: F401

__version__ = "1.0.0-shim"
