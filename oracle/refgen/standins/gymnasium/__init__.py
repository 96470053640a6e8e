"""Minimal stand-in for `gymnasium` (absent here): Env, spaces.Box, register/make/registry and
the TimeLimit semantics (truncated = elapsed_steps >= max_episode_steps). Test infrastructure only."""
import importlib
from . import spaces  # noqa
from .core import Env  # noqa
from .envs.registration import register, registry, make  # noqa
