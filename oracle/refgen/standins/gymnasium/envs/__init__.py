from . import registration  # noqa
from .registration import registry  # noqa  (utils/utils.py:248 walks gym.envs.registry)
