"""ctypes binding of the CPU oracle (oracle/libca_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/ca_oracle.h for the parity status).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libca_oracle.so")
# CA_ORACLE_SANITIZED=1 (tests/test_oracle_sanitized.py, a child interpreter started under LD_PRELOAD=libasan): load the
# -fsanitize=address,undefined build of the same source (make -C oracle asan) instead
_SANITIZED = os.environ.get("CA_ORACLE_SANITIZED") == "1"
if _SANITIZED:
    _LIB_PATH = os.path.join(_HERE, "libca_oracle_asan.so")

OBS_DIM = 64
DONE_XLESS, DONE_GOAL, DONE_REGOAL = 0, 1, 2
F_OBS, F_STATS, F_AUTORESET, F_NODONE, F_FREEZE = 1, 2, 4, 8, 16
PREC_F32, PREC_F64 = 0, 1
SCN_CROWD, SCN_CIRCLE, SCN_DOORWAY, SCN_CONGESTED, SCN_INCOMING, SCN_BLOCKS, SCN_DEADLOCK, SCN_CROWD_SEPARATED = range(8)

(FLD_POS_X, FLD_POS_Y, FLD_VEL_X, FLD_VEL_Y, FLD_PREF_X, FLD_PREF_Y, FLD_GOAL_X, FLD_GOAL_Y,
 FLD_GOAL2_X, FLD_GOAL2_Y, FLD_REWARD, FLD_AGENT_DONE, FLD_ARRIVE_STEP, FLD_NB_COUNT, FLD_NB_IDX,
 FLD_OBST_COUNT, FLD_OBST_IDX, FLD_OBS, FLD_OBS64, FLD_REWARD64, FLD_STEP_COUNT, FLD_ARENA_DONE,
 FLD_EPISODE, FLD_REGOAL_COUNT, FLD_ALAN_WEIGHTS, FLD_ALAN_TIMES, FLD_ALAN_ACTION, FLD_ARENA_STATS,
 FLD_OBS_MARGIN) = range(29)


class Config(C.Structure):
    _fields_ = [
        ("n_arenas", C.c_int32), ("n_agents", C.c_int32), ("arena_offset", C.c_int64),
        ("seed", C.c_uint64), ("reward_scale", C.c_double), ("time_step", C.c_float), ("neighbor_dist", C.c_float),
        ("max_neighbors", C.c_int32), ("time_horizon", C.c_float), ("time_horizon_obst", C.c_float),
        ("radius", C.c_float), ("max_speed", C.c_float), ("max_obst_neighbors", C.c_int32),
        ("max_step", C.c_int32), ("done_mode", C.c_int32), ("done_x_thresh", C.c_float),
        ("spawn_x0", C.c_float), ("spawn_x1", C.c_float),
        ("spawn_y0", C.c_float), ("spawn_y1", C.c_float), ("goal_x0", C.c_float),
        ("goal_x1", C.c_float), ("goal_y0", C.c_float), ("goal_y1", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [("agent_steps", C.c_uint64), ("episodes", C.c_uint64), ("collisions", C.c_uint64),
                ("obst_collisions", C.c_uint64), ("goals_reached", C.c_uint64),
                ("obst_overflow", C.c_uint64), ("sum_reward", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    """Compile oracle/libca_oracle.so with the committed Makefile (gcc only)."""
    src = [os.path.join(_HERE, f) for f in ("ca_oracle.cpp", "ca_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "asan" if _SANITIZED else "libca_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, u32, f32, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_float, C.c_size_t
    L.orc_env_create.restype = vp
    L.orc_env_create.argtypes = [C.POINTER(Config)]
    L.orc_env_destroy.argtypes = [vp]
    L.orc_env_set_obstacles.argtypes = [vp, vp, vp, i32]
    L.orc_env_set_obstacles_per_arena.argtypes = [vp, vp, vp, vp]
    L.orc_env_obstacle_table_arena.argtypes = [vp, i32] + [vp] * 7 + [i32]
    L.orc_env_init_scenario.argtypes = [vp, i32]
    L.orc_env_set.argtypes = [vp, i32, vp, sz]
    L.orc_env_get.argtypes = [vp, i32, vp, sz]
    L.orc_env_reset.argtypes = [vp, vp, vp, u32, i32]
    L.orc_env_reset_masked.argtypes = [vp, vp, u32, i32]
    L.orc_env_step_mt.argtypes = [vp, vp, u32, i32, i32]
    L.orc_env_step.argtypes = [vp, vp, u32, i32]
    L.orc_env_orca_step.argtypes = [vp, u32, i32]
    L.orc_env_rollout.argtypes = [vp, i32, u32, i32]
    L.orc_env_rollout_mt.argtypes = [vp, vp, i32, i32, u32, i32, i32]
    L.orc_env_stats.argtypes = [vp, C.POINTER(Stats)]
    L.orc_env_alan_configure.argtypes = [vp, vp, i32, C.c_double, C.c_double, C.c_double]
    L.orc_env_alan_step.argtypes = [vp, vp, u32, i32]
    L.orc_exp64.argtypes = [C.c_double]
    L.orc_exp64.restype = C.c_double
    L.orc_env_obstacle_table.argtypes = [vp] + [vp] * 7 + [i32]
    L.orc_sim_create.restype = vp
    L.orc_sim_create.argtypes = [f32, f32, i32, f32, f32, f32, f32, f32, f32]
    L.orc_sim_destroy.argtypes = [vp]
    L.orc_sim_add_agent.argtypes = [vp, f32, f32, f32, i32, f32, f32, f32, f32, f32, f32]
    L.orc_sim_add_obstacle.argtypes = [vp, vp, i32]
    L.orc_sim_process_obstacles.argtypes = [vp]
    L.orc_sim_do_step.argtypes = [vp]
    L.orc_sim_num_agents.argtypes = [vp]
    L.orc_sim_get_agent.argtypes = [vp, i32, i32, vp]
    L.orc_sim_set_agent.argtypes = [vp, i32, i32, f32, f32]
    for n in ("num_agent_neighbors", "num_obstacle_neighbors", "next_obstacle_vertex",
              "prev_obstacle_vertex"):
        getattr(L, "orc_sim_" + n).argtypes = [vp, i32]
    L.orc_sim_agent_neighbor.argtypes = [vp, i32, i32]
    L.orc_sim_obstacle_neighbor.argtypes = [vp, i32, i32]
    L.orc_sim_obstacle_vertex.argtypes = [vp, i32, vp]
    L.orc_sim_num_obstacle_vertices.argtypes = [vp]
    L.orc_line_intersection_f64.argtypes = [vp] * 5
    L.orc_comp_laser_f64.argtypes = [vp, vp, i32, vp, vp]
    L.orc_comp_laser_f32.argtypes = [vp, vp, i32, vp, vp]
    L.orc_sincos64.argtypes = [C.c_double, vp, vp]
    L.orc_pref_dir64.argtypes = [f32, f32, C.c_double, C.c_double, vp]
    L.orc_philox4x32.argtypes = [u32] * 6 + [vp]
    L.orc_debug_branch_name.restype = C.c_char_p
    L.orc_debug_branch_name.argtypes = [i32]
    L.orc_debug_branches.argtypes = [vp, i32]
    L.orc_debug_capture.argtypes = [i32, i32]
    L.orc_debug_captured.argtypes = [vp, vp, vp, i32, vp]
    L.orc_ray_table.argtypes = [C.c_double, vp]
    L.orc_octagon_table.argtypes = [C.c_double, vp]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_config(**kw):
    """Defaults = the reference env (env.py:26-44,130,359,396,478)."""
    d = dict(n_arenas=1, n_agents=10, arena_offset=0, seed=0, time_step=1 / 60., neighbor_dist=1.5,
             max_neighbors=5, time_horizon=1.5, time_horizon_obst=1.5, radius=0.5, max_speed=1.0,
             max_obst_neighbors=8, max_step=1000, done_mode=DONE_XLESS, done_x_thresh=2.0,
             reward_scale=0.3, spawn_x0=5.0, spawn_x1=10.0, spawn_y0=0.0, spawn_y1=10.0,
             goal_x0=0.0, goal_x1=10.0, goal_y0=0.0, goal_y1=10.0)
    d.update(kw)
    return Config(**d)


_FIELD_SHAPES = {
    FLD_NB_IDX: ("K", np.int32), FLD_OBST_IDX: ("S", np.int32), FLD_OBS: (OBS_DIM, np.float32),
    FLD_OBS64: (OBS_DIM, np.float64), FLD_REWARD64: (None, np.float64), FLD_OBS_MARGIN: (16, np.float64),
    FLD_AGENT_DONE: (None, np.int32), FLD_ARRIVE_STEP: (None, np.int32),
    FLD_NB_COUNT: (None, np.int32), FLD_OBST_COUNT: (None, np.int32),
    FLD_REGOAL_COUNT: (None, np.int32), FLD_ALAN_ACTION: (None, np.int32),
    FLD_GOAL_X: (None, np.float64), FLD_GOAL_Y: (None, np.float64), FLD_GOAL2_X: (None, np.float64),
    FLD_GOAL2_Y: (None, np.float64),
    FLD_ALAN_WEIGHTS: ("nA", np.float64), FLD_ALAN_TIMES: ("nA", np.float64),
}
_ARENA_FIELDS = (FLD_STEP_COUNT, FLD_ARENA_DONE, FLD_EPISODE)


class OracleEnv:
    def __init__(self, cfg):
        self.cfg = cfg
        self.L = lib()
        self.h = self.L.orc_env_create(C.byref(cfg))
        if not self.h:
            raise RuntimeError("orc_env_create failed")
        self.A, self.N = cfg.n_arenas, cfg.n_agents

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_env_destroy(self.h)
            self.h = None

    def _shape_dtype(self, field):
        if field == FLD_ARENA_STATS:
            return (self.A, 8), np.uint64
        if field in _ARENA_FIELDS:
            return (self.A,), np.int32
        last, dt = _FIELD_SHAPES.get(field, (None, np.float32))
        if last == "K":
            last = self.cfg.max_neighbors
        elif last == "S":
            last = self.cfg.max_obst_neighbors
        elif last == "nA":
            last = self.n_actions
        return ((self.A, self.N) if last is None else (self.A, self.N, last)), dt

    def set_obstacles(self, polys):
        verts = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in polys])
                                     if polys else np.zeros((0, 2), np.float32))
        sizes = np.asarray([len(p) for p in polys], np.int32)
        rc = self.L.orc_env_set_obstacles(self.h, _ptr(verts), _ptr(sizes), len(polys))
        assert rc == 0

    def set_obstacles_per_arena(self, worlds):
        """worlds[a] = the polygons of arena a."""
        assert len(worlds) == self.A
        polys = [np.asarray(q, np.float32).reshape(-1, 2) for w in worlds for q in w]
        verts = np.ascontiguousarray(np.concatenate(polys) if polys else np.zeros((0, 2), np.float32))
        sizes = np.asarray([len(q) for q in polys], np.int32)
        counts = np.asarray([len(w) for w in worlds], np.int32)
        assert self.L.orc_env_set_obstacles_per_arena(self.h, _ptr(verts), _ptr(sizes), _ptr(counts)) == 0

    def obstacle_table(self, cap=256, arena=None):
        px, py, ux, uy = (np.zeros(cap, np.float32) for _ in range(4))
        nx, pv, cv = (np.zeros(cap, np.int32) for _ in range(3))
        if arena is None:
            n = self.L.orc_env_obstacle_table(self.h, _ptr(px), _ptr(py), _ptr(ux), _ptr(uy), _ptr(nx),
                                              _ptr(pv), _ptr(cv), cap)
        else:
            n = self.L.orc_env_obstacle_table_arena(self.h, int(arena), _ptr(px), _ptr(py), _ptr(ux), _ptr(uy),
                                                    _ptr(nx), _ptr(pv), _ptr(cv), cap)
        return dict(px=px[:n], py=py[:n], ux=ux[:n], uy=uy[:n], next=nx[:n], prev=pv[:n], convex=cv[:n])

    def init_scenario(self, scenario):
        assert self.L.orc_env_init_scenario(self.h, scenario) == 0

    def get(self, field):
        shape, dt = self._shape_dtype(field)
        out = np.empty(shape, dt)
        rc = self.L.orc_env_get(self.h, field, _ptr(out), out.nbytes)
        assert rc == 0, (field, rc)
        return out

    def set(self, field, arr):
        shape, dt = self._shape_dtype(field)
        a = np.ascontiguousarray(np.asarray(arr, dt).reshape(shape))
        rc = self.L.orc_env_set(self.h, field, _ptr(a), a.nbytes)
        assert rc == 0, (field, rc)

    def reset(self, pos_x=None, pos_y=None, flags=F_OBS, prec=PREC_F32):
        if pos_x is None:
            rc = self.L.orc_env_reset(self.h, None, None, flags, prec)
        else:
            px = np.ascontiguousarray(np.asarray(pos_x, np.float32).reshape(self.A, self.N))
            py = np.ascontiguousarray(np.asarray(pos_y, np.float32).reshape(self.A, self.N))
            rc = self.L.orc_env_reset(self.h, _ptr(px), _ptr(py), flags, prec)
        assert rc == 0

    def reset_masked(self, mask, flags=F_OBS, prec=PREC_F32):
        m = np.ascontiguousarray(np.asarray(mask, np.int32).reshape(self.A))
        assert self.L.orc_env_reset_masked(self.h, _ptr(m), flags, prec) == 0

    def step(self, actions, flags=F_OBS, prec=PREC_F32):
        a = np.ascontiguousarray(np.asarray(actions, np.float32).reshape(self.A, self.N))
        assert self.L.orc_env_step(self.h, _ptr(a), flags, prec) == 0

    def step_mt(self, actions, flags=F_OBS, prec=PREC_F32, n_threads=1):
        """step (actions given) or orca_step (actions None) on n_threads host threads."""
        if actions is None:
            assert self.L.orc_env_step_mt(self.h, None, flags, prec, n_threads) == 0
        else:
            a = np.ascontiguousarray(np.asarray(actions, np.float32).reshape(self.A, self.N))
            assert self.L.orc_env_step_mt(self.h, _ptr(a), flags, prec, n_threads) == 0

    def orca_step(self, flags=0, prec=PREC_F32):
        assert self.L.orc_env_orca_step(self.h, flags, prec) == 0

    def rollout(self, steps, flags=0, n_threads=1):
        assert self.L.orc_env_rollout(self.h, steps, flags, n_threads) == 0

    def rollout_mt(self, steps, actions_pool=None, flags=F_OBS, prec=PREC_F32, n_threads=1):
        """The multi-core CPU baseline: every thread steps its own block of arenas through all `steps` steps (no
        per-step join).  actions_pool: None (ORCA-only) or [pool, A, N]; step s uses entry s % pool."""
        if actions_pool is None:
            assert self.L.orc_env_rollout_mt(self.h, None, 0, steps, flags, prec, n_threads) == 0
        else:
            a = np.ascontiguousarray(np.asarray(actions_pool, np.float32).reshape(-1, self.A, self.N))
            assert self.L.orc_env_rollout_mt(self.h, _ptr(a), a.shape[0], steps, flags, prec, n_threads) == 0

    def alan_configure(self, actions, temp=0.2, timewindow=2.0, time_step=1 / 60.):
        a = np.ascontiguousarray(np.asarray(actions, np.float64).reshape(-1, 2))
        self.n_actions = a.shape[0]
        assert self.L.orc_env_alan_configure(self.h, _ptr(a), a.shape[0], temp, timewindow, time_step) == 0

    def alan_step(self, u=None, flags=0, prec=PREC_F32):
        if u is None:
            assert self.L.orc_env_alan_step(self.h, None, flags, prec) == 0
        else:
            uu = np.ascontiguousarray(np.asarray(u, np.float64).reshape(self.A, self.N))
            assert self.L.orc_env_alan_step(self.h, _ptr(uu), flags, prec) == 0

    def stats(self):
        s = Stats()
        self.L.orc_env_stats(self.h, C.byref(s))
        return s.as_dict()

    def state(self):
        """The trajectory-relevant state as a dict of arrays (used by parity tests)."""
        names = dict(pos_x=FLD_POS_X, pos_y=FLD_POS_Y, vel_x=FLD_VEL_X, vel_y=FLD_VEL_Y,
                     pref_x=FLD_PREF_X, pref_y=FLD_PREF_Y, goal_x=FLD_GOAL_X, goal_y=FLD_GOAL_Y,
                     agent_done=FLD_AGENT_DONE, step_count=FLD_STEP_COUNT, arena_done=FLD_ARENA_DONE)
        return {k: self.get(v) for k, v in names.items()}


class OracleSim:
    """Thin object over the single-simulator entry points (one arena, per-scalar getters)."""

    def __init__(self, time_step, neighbor_dist, max_neighbors, time_horizon, time_horizon_obst,
                 radius, max_speed, velocity=(0.0, 0.0)):
        self.L = lib()
        self.h = self.L.orc_sim_create(time_step, neighbor_dist, max_neighbors, time_horizon,
                                       time_horizon_obst, radius, max_speed, velocity[0], velocity[1])

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_sim_destroy(self.h)
            self.h = None

    def _get2(self, i, what):
        out = (C.c_float * 2)()
        self.L.orc_sim_get_agent(self.h, i, what, out)
        return (out[0], out[1])


# --- stand-alone helpers -------------------------------------------------------------------------
def line_intersection(L1, L2):
    a = np.asarray([L1[0][0], L1[0][1], L1[1][0], L1[1][1]], np.float64)
    b = np.asarray([L2[0][0], L2[0][1], L2[1][0], L2[1][1]], np.float64)
    d, x, y = C.c_double(), C.c_double(), C.c_double()
    lib().orc_line_intersection_f64(_ptr(a), _ptr(b), C.byref(d), C.byref(x), C.byref(y))
    return d.value, (x.value, y.value)


def comp_laser(ray_ends, segs, orientation, dtype=np.float64):
    """ray_ends [16,2]; segs [m,6] = (x1,y1,x2,y2,vx,vy); returns [16,4]."""
    r = np.ascontiguousarray(ray_ends, dtype)
    s = np.ascontiguousarray(segs, dtype).reshape(-1, 6)
    o = np.ascontiguousarray(orientation, dtype)
    out = np.zeros((16, 4), dtype)
    f = lib().orc_comp_laser_f64 if dtype == np.float64 else lib().orc_comp_laser_f32
    f(_ptr(r), _ptr(s), s.shape[0], _ptr(o), _ptr(out))
    return out


def sincos64(a):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos64(float(a), C.byref(s), C.byref(c))
    return s.value, c.value


def exp64(x):
    return lib().orc_exp64(float(x))


def pref_dir64(px, py, gx, gy):
    out = (C.c_double * 2)()
    lib().orc_pref_dir64(px, py, gx, gy, out)
    return out[0], out[1]


def philox4x32(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out)
    return tuple(out)


def ray_table(neighbor_dist):
    out = np.zeros(32, np.float64)
    lib().orc_ray_table(float(neighbor_dist), _ptr(out))
    return out.reshape(16, 2)


def octagon_table(radius):
    out = np.zeros(32, np.float64)
    lib().orc_octagon_table(float(radius), _ptr(out))
    return out.reshape(8, 4)


# ---- diagnostics (tests/test_oracle_orca_definition.py): branch counters and captured ORCA lines of the calling thread ----
def branch_names():
    L = lib()
    return [L.orc_debug_branch_name(k).decode() for k in range(L.orc_debug_branch_count())]


def branch_counts(reset=False):
    """{branch name: times taken} on this thread since the last reset (SURVEY App. A.3 / A.4 / A.5 branch by branch)."""
    L = lib()
    n = L.orc_debug_branch_count()
    out = np.zeros(n, np.uint64)
    L.orc_debug_branches(_ptr(out), n)
    if reset:
        L.orc_debug_branches_reset()
    return dict(zip(branch_names(), (int(v) for v in out)))


def capture_next(arena, agent):
    """Record the ORCA lines of `agent` of `arena` during the next serial step on this thread (a PyRVOSimulator is arena 0)."""
    lib().orc_debug_capture(int(arena), int(agent))


def captured(cap=64):
    """dict(lines [n,4] = point.x, point.y, dir.x, dir.y (obstacle lines first), n_obst_lines, line_edge [n_obst_lines],
    nb_edge / nb_branch per obstacle neighbour in list order, fail = the line LP2 failed at (n: feasible)), or None."""
    L = lib()
    lines = np.zeros((cap, 4), np.float32)
    edge, tag, info = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(4, np.int32)
    n = L.orc_debug_captured(_ptr(lines), _ptr(edge), _ptr(tag), cap, _ptr(info))
    if n < 0:
        return None
    names = branch_names()
    return dict(lines=lines[:n].copy(), n_obst_lines=int(info[0]), line_edge=edge[:info[0]].copy(),
                nb_edge=(tag[:info[1]] >> 8), nb_branch=[names[t & 0xFF] for t in tag[:info[1]]], fail=int(info[2]))
