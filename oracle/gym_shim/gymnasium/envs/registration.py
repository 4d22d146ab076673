"""gymnasium.envs.registration stand-in: id -> entry_point + kwargs, and make()."""
from __future__ import annotations

import importlib
from dataclasses import dataclass, field
from typing import Any

from ..error import NameNotFound


@dataclass
class EnvSpec:
    id: str
    entry_point: Any = None
    kwargs: dict = field(default_factory=dict)
    max_episode_steps: Any = None
    reward_threshold: Any = None
    nondeterministic: bool = False


registry: dict[str, EnvSpec] = {}


def register(id, entry_point=None, kwargs=None, **extra):
    registry[id] = EnvSpec(id=id, entry_point=entry_point, kwargs=dict(kwargs or {}))


def _load(entry_point):
    if callable(entry_point):
        return entry_point
    mod_name, attr = entry_point.split(":")
    return getattr(importlib.import_module(mod_name), attr)


def make(id, **kwargs):
    if isinstance(id, EnvSpec):
        spec = id
    else:
        if id not in registry:
            raise NameNotFound(id)
        spec = registry[id]
    kw = dict(spec.kwargs)
    kw.update(kwargs)
    env = _load(spec.entry_point)(**kw)
    env.spec = spec
    return env
