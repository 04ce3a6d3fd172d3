"""Shared helpers for the parity tests: build the HIP environment and the CPU oracle in the same
configuration and compare them field by field."""
import numpy as np

from collision_avoidance_amd import scenarios
from oracle import oracle as o

SCN = {"crowd": o.SCN_CROWD, "circle": o.SCN_CIRCLE, "doorway": o.SCN_DOORWAY, "congested": o.SCN_CONGESTED,
       "incoming": o.SCN_INCOMING, "blocks": o.SCN_BLOCKS, "deadlock": o.SCN_DEADLOCK,
       "crowd_separated": o.SCN_CROWD_SEPARATED}


def scenario_params(scenario, n_agents, **over):
    if scenario == "doorway":
        p = scenarios.env_params()
    else:
        p = scenarios.alan_params(n_agents, scenario)
    p.update(over)
    return p


def make_oracle(A, N, scenario, params, seed=0, arena_offset=0, max_obst_neighbors=None, polys=None):
    """polys: None = the scenario's own obstacles (a world per arena where the scenario draws them at random),
    a list of polygons = the same for every arena, dict(per_arena=[...]) = explicit per-arena worlds."""
    worlds = None
    if polys is None:
        worlds = scenarios.obstacle_worlds(scenario, A, N, params["radius"], seed, arena_offset)
        polys = scenarios.obstacles(scenario, N, params["radius"]) if worlds is None else []
    elif isinstance(polys, dict):
        worlds, polys = list(polys["per_arena"]), []
    n_edges = max(sum(len(q) for q in w) for w in worlds) if worlds else sum(len(q) for q in polys)
    S = max(1, min(16, n_edges)) if max_obst_neighbors is None else max_obst_neighbors
    cfg = o.make_config(n_arenas=A, n_agents=N, seed=seed, arena_offset=arena_offset,
                        max_obst_neighbors=S, **params)
    env = o.OracleEnv(cfg)
    if worlds is not None:
        env.set_obstacles_per_arena(worlds)
    else:
        env.set_obstacles(polys)
    env.init_scenario(SCN[scenario])
    return env


def make_gpu(A, N, scenario, params, seed=0, arena_offset=0, max_obst_neighbors=None, use_torch=False,
             polys=None, **kw):
    from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
    return VecCollisionAvoidanceEnv(A, N, scenario=scenario, params=params, seed=seed,
                                    arena_offset=arena_offset, max_obst_neighbors=max_obst_neighbors,
                                    use_torch=use_torch,
                                    obstacles="scenario" if polys is None else polys, **kw)


def _eq(a, b, what):
    """Bit patterns: +0.0 is not -0.0, a NaN equals only the same NaN."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype.kind == b.dtype.kind, (what, a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        assert a.dtype == b.dtype, (what, a.dtype, b.dtype)
        u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
        a, b = a.view(u), b.view(u)
    ok = np.array_equal(a, b)
    if not ok:
        bad = np.argwhere(a != b)
        first = tuple(bad[0])
        raise AssertionError("%s differs at %d of %d entries; first %s: gpu=%r oracle=%r (bit patterns if float)"
                             % (what, len(bad), a.size, first, a[first], b[first]))


def assert_state_equal(gpu, orc, what="", obs=False, reward=False, lists=True):
    """Bit-exact comparison of the HIP environment with the oracle (ORC_PREC_F32)."""
    from collision_avoidance_amd import _lib
    pairs = [("pos_x", _lib.FLD_POS_X, o.FLD_POS_X), ("pos_y", _lib.FLD_POS_Y, o.FLD_POS_Y),
             ("vel_x", _lib.FLD_VEL_X, o.FLD_VEL_X), ("vel_y", _lib.FLD_VEL_Y, o.FLD_VEL_Y),
             ("pref_x", _lib.FLD_PREF_X, o.FLD_PREF_X), ("pref_y", _lib.FLD_PREF_Y, o.FLD_PREF_Y),
             ("goal_x", _lib.FLD_GOAL_X, o.FLD_GOAL_X), ("goal_y", _lib.FLD_GOAL_Y, o.FLD_GOAL_Y),
             ("agent_done", _lib.FLD_AGENT_DONE, o.FLD_AGENT_DONE),
             ("step_count", _lib.FLD_STEP_COUNT, o.FLD_STEP_COUNT),
             ("arena_done", _lib.FLD_ARENA_DONE, o.FLD_ARENA_DONE),
             ("episode", _lib.FLD_EPISODE, o.FLD_EPISODE),
             ("regoal_count", _lib.FLD_REGOAL_COUNT, o.FLD_REGOAL_COUNT)]
    for name, gf, of in pairs:
        _eq(gpu.get(gf), orc.get(of), "%s %s" % (what, name))
    if lists:
        gc, gi = gpu.neighbor_lists()
        oc, oi = orc.get(o.FLD_NB_COUNT), orc.get(o.FLD_NB_IDX)
        _eq(gc, oc, what + " nb_count")
        K = oi.shape[2]
        mask = np.arange(K)[None, None, :] < oc[:, :, None]
        _eq(np.where(mask, gi, -1), np.where(mask, oi, -1), what + " nb_idx")
        gc, gi = gpu.obstacle_neighbor_lists()
        oc, oi = orc.get(o.FLD_OBST_COUNT), orc.get(o.FLD_OBST_IDX)
        _eq(gc, oc, what + " obst_count")
        S = oi.shape[2]
        mask = np.arange(S)[None, None, :] < oc[:, :, None]
        _eq(np.where(mask, gi, -1), np.where(mask, oi, -1), what + " obst_idx")
    if obs:
        _eq(gpu.get(_lib.FLD_OBS), orc.get(o.FLD_OBS), what + " obs")
    if reward:
        _eq(gpu.get(_lib.FLD_REWARD), orc.get(o.FLD_REWARD), what + " reward")


def assert_stats_equal(gpu, orc, what=""):
    g, s = gpu.stats(), orc.stats()
    for k in ("agent_steps", "episodes", "collisions", "obst_collisions", "goals_reached", "obst_overflow"):
        assert g[k] == s[k], "%s stats.%s gpu=%d oracle=%d" % (what, k, g[k], s[k])
    assert abs(g["sum_reward"] - s["sum_reward"]) <= 1e-9 * max(1.0, abs(s["sum_reward"])), (g, s)


class OracleVec(object):
    """The subset of VecCollisionAvoidanceEnv's interface that the adapters use, backed by the CPU oracle:
    lets the host-side adapter logic be tested without a GPU and gives the GPU tests their reference."""
    use_torch = False

    def __init__(self, A, N, scenario, params, seed=0):
        self.env = make_oracle(A, N, scenario, params, seed=seed)
        self.A, self.N, self.cfg = A, N, self.env.cfg

    def _fld(self, field):
        from collision_avoidance_amd import _lib
        name = [k for k in dir(_lib) if k.startswith("FLD_") and getattr(_lib, k) == field][0]
        return getattr(o, name)

    def get(self, field):
        return self.env.get(self._fld(field))

    def reset(self, with_obs=True):
        self.env.reset(flags=o.F_OBS if with_obs else 0)
        return self.env.get(o.FLD_OBS) if with_obs else None

    def reset_masked(self, mask, with_obs=True):
        self.env.reset_masked(mask, flags=o.F_OBS if with_obs else 0)
        return self.env.get(o.FLD_OBS) if with_obs else None

    def step(self, actions, with_obs=True, stats=False, autoreset=False):
        flags = (o.F_OBS if with_obs else 0) | (o.F_STATS if stats else 0) | (o.F_AUTORESET if autoreset else 0)
        self.env.step(actions, flags=flags)
        return (self.env.get(o.FLD_OBS) if with_obs else None), self.env.get(o.FLD_REWARD), \
            self.env.get(o.FLD_ARENA_DONE), {}

    def arena_stats(self):
        r = self.env.get(o.FLD_ARENA_STATS)
        return dict(episodes=r[:, 0], collisions=r[:, 1], obst_collisions=r[:, 2], goals_reached=r[:, 3],
                    obst_overflow=r[:, 4], sum_reward=r[:, 5].copy().view(np.float64), frozen_steps=r[:, 6],
                    last_episode_steps=(r[:, 7] >> np.uint64(32)).astype(np.int64),
                    last_episode_arrived=(r[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64))

    def close(self):
        pass


def _lib_fld(name):
    from collision_avoidance_amd import _lib
    return getattr(_lib, "FLD_" + name)
