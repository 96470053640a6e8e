import importlib

from ..core import Env

registry = {}


class _Spec:
    def __init__(self, id, entry_point, max_episode_steps):
        self.id, self.entry_point, self.max_episode_steps = id, entry_point, max_episode_steps


def register(id, entry_point, max_episode_steps=None, **kw):
    registry[id] = _Spec(id, entry_point, max_episode_steps)


class TimeLimit(Env):
    """gymnasium>=0.29 TimeLimit (a gymnasium.Wrapper, i.e. an Env: algs/iwpg/iwpg.py:222 asserts it):
    truncated = elapsed_steps >= max_episode_steps."""

    def __init__(self, env, max_episode_steps):
        self.env = env
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0

    @property
    def unwrapped(self):
        return self.env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kw):
        self._elapsed_steps = 0
        return self.env.reset(**kw)

    def step(self, action):
        obs, r, term, trunc, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            trunc = True
        return obs, r, term, trunc, info


def make(id, **kwargs):
    spec = registry[id]
    mod, cls = spec.entry_point.split(':')
    env = getattr(importlib.import_module(mod), cls)(**kwargs)
    if spec.max_episode_steps:
        env = TimeLimit(env, spec.max_episode_steps)
    return env
