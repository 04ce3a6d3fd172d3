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


try:  # the reference registers a gym id (collision_avoidance/__init__.py:3-6)
    from gym.envs.registration import register as _register
    _register(id='collision_avoidance-v0', entry_point='collision_avoidance_amd.envs:Collision_Avoidance_Env')
except Exception:  # gym absent or id already registered
    pass
