"""ctypes binding of libcaenv.so (the C ABI of include/ca_env.h).

Fails loudly: if the library is missing or a call returns an error code a RuntimeError carrying
ca_last_error() is raised.  There is no CPU fallback of any kind.
"""
import ctypes as C
import os

from . import build as _build

OBS_DIM = 64
MAX_NEIGHBORS, MAX_OBST_NEIGHBORS, MAX_AGENTS = 16, 16, 1024
DONE_XLESS, DONE_GOAL, DONE_REGOAL = 0, 1, 2
F_OBS, F_STATS, F_AUTORESET, F_NODONE, F_FREEZE = 1, 2, 4, 8, 16
SCN_CROWD, SCN_CIRCLE, SCN_DOORWAY, SCN_CONGESTED, SCN_INCOMING, SCN_BLOCKS, SCN_DEADLOCK, SCN_CROWD_SEPARATED = range(8)

(FLD_POS_X, FLD_POS_Y, FLD_VEL_X, FLD_VEL_Y, FLD_PREF_X, FLD_PREF_Y, FLD_GOAL_X, FLD_GOAL_Y,
 FLD_GOAL2_X, FLD_GOAL2_Y, FLD_REWARD, FLD_AGENT_DONE, FLD_ARRIVE_STEP, FLD_NB_COUNT, FLD_NB_IDX,
 FLD_OBST_COUNT, FLD_OBST_IDX, FLD_OBS, FLD_STEP_COUNT, FLD_ARENA_DONE, FLD_EPISODE,
 FLD_REGOAL_COUNT, FLD_ALAN_WEIGHTS, FLD_ALAN_TIMES, FLD_ALAN_ACTION, FLD_ARENA_STATS) = range(26)

EXPORTS = ("ca_create", "ca_destroy", "ca_last_error", "ca_set_stream", "ca_set_obstacles", "ca_init_scenario", "ca_set",
           "ca_get", "ca_field_ptr", "ca_bind_obs", "ca_reset", "ca_step", "ca_step_host", "ca_orca_step", "ca_observe", "ca_rollout",
           "ca_get_stats", "ca_reset_stats", "ca_sync", "ca_debug_math", "ca_profile", "ca_profile_read", "ca_launch_info",
           "ca_alan_configure", "ca_alan_step", "ca_alan_rollout", "ca_reset_masked", "ca_get_obstacles",
           "ca_set_obstacles_per_arena", "ca_get_obstacles_arena", "ca_solver_info", "ca_source_sha", "ca_host_alloc", "ca_host_free",
           "ca_step_packed", "ca_allow_obstacle_overflow")


class Config(C.Structure):
    """struct ca_config (include/ca_env.h)."""
    _fields_ = [
        ("n_arenas", C.c_int32), ("n_agents", C.c_int32), ("arena_offset", C.c_int64),
        ("seed", C.c_uint64), ("reward_scale", C.c_double), ("time_step", C.c_float),
        ("neighbor_dist", C.c_float), ("max_neighbors", C.c_int32), ("time_horizon", C.c_float),
        ("time_horizon_obst", C.c_float), ("radius", C.c_float), ("max_speed", C.c_float),
        ("max_obst_neighbors", C.c_int32), ("max_step", C.c_int32), ("done_mode", C.c_int32),
        ("done_x_thresh", C.c_float), ("spawn_x0", C.c_float), ("spawn_x1", C.c_float),
        ("spawn_y0", C.c_float), ("spawn_y1", C.c_float), ("goal_x0", C.c_float),
        ("goal_x1", C.c_float), ("goal_y0", C.c_float), ("goal_y1", C.c_float),
    ]


class Stats(C.Structure):
    """struct ca_stats (include/ca_env.h)."""
    _fields_ = [("agent_steps", C.c_uint64), ("episodes", C.c_uint64), ("collisions", C.c_uint64),
                ("obst_collisions", C.c_uint64), ("goals_reached", C.c_uint64),
                ("obst_overflow", C.c_uint64), ("sum_reward", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib_path():
    return _build.LIB_PATH


def load():
    """Load libcaenv.so (must have been built: __graft_entry__.build() / python -m
    collision_avoidance_amd.build).  Raises if it is absent -- nothing else can run the path."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("libcaenv.so is not built (%s); run `python -m collision_avoidance_amd.build` "
                           "-- the HIP extension is the only implementation of this path" % path)
    L = C.CDLL(path)
    vp, i32, u32, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t
    L.ca_create.argtypes = [C.POINTER(Config), C.c_int, vp, C.POINTER(vp)]
    L.ca_destroy.argtypes = [vp]
    L.ca_last_error.argtypes = [vp]
    L.ca_last_error.restype = C.c_char_p
    L.ca_set_stream.argtypes = [vp, vp]
    L.ca_set_obstacles.argtypes = [vp, vp, vp, i32]
    L.ca_init_scenario.argtypes = [vp, i32]
    L.ca_get_obstacles.argtypes = [vp, vp, vp, vp, i32, C.POINTER(i32)]
    L.ca_set_obstacles_per_arena.argtypes = [vp, vp, vp, vp]
    L.ca_get_obstacles_arena.argtypes = [vp, i32, vp, vp, vp, i32, C.POINTER(i32)]
    L.ca_set.argtypes = [vp, i32, vp, sz, i32]
    L.ca_get.argtypes = [vp, i32, vp, sz, i32]
    L.ca_field_ptr.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz)]
    L.ca_bind_obs.argtypes = [vp, vp, sz]
    L.ca_reset.argtypes = [vp, vp, vp, i32, u32]
    L.ca_reset_masked.argtypes = [vp, vp, i32, u32]
    L.ca_step.argtypes = [vp, vp, u32]
    L.ca_step_host.argtypes = [vp, vp, u32]
    L.ca_step_packed.argtypes = [vp, vp, u32, vp, sz]
    L.ca_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.ca_host_free.argtypes = [vp, vp]
    L.ca_orca_step.argtypes = [vp, u32]
    L.ca_observe.argtypes = [vp]
    L.ca_rollout.argtypes = [vp, i32, u32]
    L.ca_alan_configure.argtypes = [vp, vp, i32, C.c_double, C.c_double, C.c_double]
    L.ca_alan_step.argtypes = [vp, vp, i32, u32]
    L.ca_alan_rollout.argtypes = [vp, i32, u32]
    L.ca_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.ca_reset_stats.argtypes = [vp]
    L.ca_sync.argtypes = [vp]
    L.ca_allow_obstacle_overflow.argtypes = [vp, i32]
    L.ca_debug_math.argtypes = [vp, i32, vp, vp, i32]
    L.ca_profile.argtypes = [vp, i32]
    L.ca_profile_read.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_float)]
    L.ca_launch_info.argtypes = [vp] + [C.POINTER(i32)] * 4
    L.ca_solver_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.ca_source_sha.argtypes = []
    L.ca_source_sha.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("ca_last_error", "ca_source_sha"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(L, handle, rc, what):
    if rc != 0:
        msg = L.ca_last_error(handle)
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
