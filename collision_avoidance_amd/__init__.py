"""MI355X-native batched collision-avoidance environment (ORCA + laser observation in HIP).

The compute path lives in libcaenv.so (collision_avoidance_amd/csrc, C ABI in include/ca_env.h);
this package is the thin host side.  Importing it does not touch the GPU.
"""
from . import scenarios  # noqa: F401

__all__ = ["scenarios", "VecCollisionAvoidanceEnv", "Collision_Avoidance_Env", "CollisionAvoidanceEnv"]


def __getattr__(name):
    if name == "VecCollisionAvoidanceEnv":
        from .vec_env import VecCollisionAvoidanceEnv
        return VecCollisionAvoidanceEnv
    if name in ("Collision_Avoidance_Env", "CollisionAvoidanceEnv"):
        from . import envs
        return getattr(envs, name)
    raise AttributeError(name)


def _register_gym_id():
    """The reference registers a gym id on import (collision_avoidance/__init__.py:3-6; run_rllib.py:82 resolves it with
    gym.make('collision_avoidance-v0')).  gym is optional here: without it nothing is registered; with it a failure to
    register anything but "already registered" (a second import path of the same package) is raised, not swallowed."""
    try:
        from gym.envs.registration import register
    except ImportError:
        return False
    try:
        register(id='collision_avoidance-v0', entry_point='collision_avoidance_amd.envs:Collision_Avoidance_Env')
    except Exception as e:       # gym.error.Error("Cannot re-register id: ...") in every gym version
        if "register" not in str(e).lower():
            raise
    return True


_register_gym_id()
