"""gymnasium.core stand-in: Env seeding semantics + the three wrapper bases."""
from __future__ import annotations

from typing import Any, TypeVar

from .utils import seeding

ObsType = TypeVar("ObsType")
ActType = TypeVar("ActType")
WrapperObsType = TypeVar("WrapperObsType")
WrapperActType = TypeVar("WrapperActType")


class Env:
    metadata: dict = {"render_modes": []}
    render_mode = None
    spec = None
    _np_random = None
    _np_random_seed = None

    def reset(self, *, seed=None, options=None):
        # gymnasium.Env.reset: (re)seed the generator only when a seed is given
        if seed is not None:
            self._np_random, self._np_random_seed = seeding.np_random(seed)

    def step(self, action):
        raise NotImplementedError

    def render(self):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, self._np_random_seed = seeding.np_random()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value
        self._np_random_seed = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self._action_space = None
        self._observation_space = None
        self._metadata = None

    def __getattr__(self, name):
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def action_space(self):
        return self.env.action_space if self._action_space is None else self._action_space

    @action_space.setter
    def action_space(self, s):
        self._action_space = s

    @property
    def observation_space(self):
        return self.env.observation_space if self._observation_space is None else self._observation_space

    @observation_space.setter
    def observation_space(self, s):
        self._observation_space = s

    @property
    def metadata(self):
        return self.env.metadata if self._metadata is None else self._metadata

    @metadata.setter
    def metadata(self, v):
        self._metadata = v

    @property
    def render_mode(self):
        return self.env.render_mode

    @property
    def np_random(self):
        return self.env.np_random

    @np_random.setter
    def np_random(self, v):
        self.env.np_random = v

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        return self.env.step(action)

    def render(self):
        return self.env.render()

    def close(self):
        return self.env.close()


class ObservationWrapper(Wrapper):
    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        return self.observation(obs), info

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return self.observation(obs), reward, terminated, truncated, info

    def observation(self, observation):
        raise NotImplementedError


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))

    def action(self, action):
        raise NotImplementedError


class RewardWrapper(Wrapper):
    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return obs, self.reward(reward), terminated, truncated, info
