"""phoenix-drone-simulation_amd -- MI355X-native batched CrazyFlie SimplePhysics environments.

Drop-in for ONE hot path of SvenGronauer/phoenix-drone-simulation: the per-env
`env.reset()/env.step()` loop of the three `*SimpleEnv-v0` ids, replaced by a lockstep HIP kernel
over N environments (csrc/pds_step.h, instantiated in csrc/pds_task_*.hip) behind the C ABI of include/pds.h.
"""
from .build import build_library, library_path  # noqa: F401
from .envs import (DroneVecEnv, DroneHoverSimpleEnv, DroneCircleSimpleEnv, DroneTakeOffSimpleEnv,  # noqa: F401
                   make, register, registry, Box)
from . import native  # noqa: F401
from .sharding import shard_range, make_sharded, all_gather_obs, P2PObsGather  # noqa: F401

__all__ = ["make", "register", "registry", "DroneVecEnv", "DroneHoverSimpleEnv",
           "DroneCircleSimpleEnv", "DroneTakeOffSimpleEnv", "Box", "build_library", "library_path",
           "native", "shard_range", "make_sharded", "all_gather_obs", "P2PObsGather"]
