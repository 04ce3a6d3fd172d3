"""Scenario geometry and parameter presets (host side, pure Python).

Everything here restates constants of the reference so that a scenario is the same world the
reference builds; citations are into /root/reference/collision_avoidance/ (env.py =
envs/collision_avoidence_env.py, ALAN = ALAN/ALAN_true.py).
"""
from math import pi, sqrt

SCN_CROWD, SCN_CIRCLE, SCN_DOORWAY, SCN_CONGESTED, SCN_INCOMING, SCN_BLOCKS, SCN_DEADLOCK, SCN_CROWD_SEPARATED = range(8)
SCENARIO_IDS = {"crowd": SCN_CROWD, "circle": SCN_CIRCLE, "doorway": SCN_DOORWAY, "congested": SCN_CONGESTED,
                "incoming": SCN_INCOMING, "blocks": SCN_BLOCKS, "deadlock": SCN_DEADLOCK,
                "crowd_separated": SCN_CROWD_SEPARATED}   # the crowd with starts >= 2 r apart (SURVEY 8d variant)
SCENARIO_NAMES = {v: k for k, v in SCENARIO_IDS.items()}

DONE_XLESS, DONE_GOAL, DONE_REGOAL = 0, 1, 2


def _rect(upper_left, upper_right, bottom_right, bottom_left):
    """env.py:143-149 / ALAN:474-480: an obstacle is the 4-vertex polygon in the given order."""
    return [upper_left, upper_right, bottom_right, bottom_left]


def crowd_envsize(n_agents, radius=0.5):
    """ALAN:272"""
    return sqrt(2 * radius * n_agents) * 2


def circle_envsize(n_agents, radius=0.5):
    """ALAN:299-301"""
    return 2 * (radius * 3 * n_agents / (2 * pi)) + 4 * radius


def envsize(scenario, n_agents, radius=0.5):
    """The world size each ALAN scenario derives from the agent count."""
    scenario = SCENARIO_NAMES.get(scenario, scenario)
    if scenario in ("crowd", "crowd_separated"):
        return crowd_envsize(n_agents, radius)
    if scenario == "circle":
        return circle_envsize(n_agents, radius)
    if scenario == "congested":
        return sqrt(2 * radius * n_agents) * 3          # ALAN:177
    if scenario in ("incoming", "deadlock"):
        return sqrt(2 * radius * n_agents) * 10         # ALAN:215, 379
    if scenario == "blocks":
        return 3 * radius * n_agents                    # ALAN:335
    if scenario == "doorway":
        return 10
    raise ValueError("unknown scenario %r" % (scenario,))


def obstacles(scenario, n_agents, radius=0.5, seed=0):
    """Obstacle polygons of a scenario as lists of (x, y).  For "blocks" this is ONE layout (numpy RandomState(seed));
    a batch draws a layout per arena instead: obstacle_worlds() / blocks_worlds()."""
    scenario = SCENARIO_NAMES.get(scenario, scenario)
    r = radius
    if scenario == "crowd_separated":
        scenario = "crowd"
    if scenario == "congested":                                            # ALAN:195-208
        e = envsize(scenario, n_agents, r)
        return [_rect((-e, 0.0), (-e, e), (e, e), (e, 0.0)),
                _rect((0.1 * e, 0.0), (0.1 * e + 0.5, 0.0), (0.1 * e + 0.5, e / 2 - 1.25 * r), (0.1 * e, e / 2 - 1.25 * r)),
                _rect((0.1 * e, e / 2 + 1.25 * r), (0.1 * e + 0.5, e / 2 + 1.25 * r), (0.1 * e + 0.5, e), (0.1 * e, e))]
    if scenario == "incoming":                                             # ALAN:263-265
        e = envsize(scenario, n_agents, r)
        return [_rect((0.0, 0.0), (0.0, e), (e, e), (e, 0.0))]
    if scenario == "blocks":                                               # ALAN:359-372
        import numpy as np
        e = envsize(scenario, n_agents, r)
        polys = [_rect((0.0, 0.0), (0.0, e), (e, e), (e, 0.0))]
        rng = np.random.RandomState(seed)
        b = e / (4 * 2)
        for _ in range(4):
            x, y = float(rng.uniform(b, e - b)), float(rng.uniform(0, e))
            polys.append(_rect((x - b / 2, y - b / 2), (x + b / 2, y - b / 2), (x + b / 2, y + b / 2), (x - b / 2, y + b / 2)))
        return polys
    if scenario == "deadlock":                                             # ALAN:418-455
        e = envsize(scenario, n_agents, r)
        lo, hi = e / 2 - 1.25 * r, e / 2 + 1.25 * r
        return [_rect((-e, 0.0), (-e, e), (2 * e, e), (2 * e, 0.0)),
                _rect((0.0, 0.0), (0.0 + 0.5, 0.0), (0.2 * e + 0.5, lo), (0.2 * e, lo)),
                _rect((0.0, e), (0.2 * e, hi), (0.2 * e + 0.5, hi), (0.0 + 0.5, e)),
                _rect((e - 0.5, 0.0), (e, 0.0), (0.8 * e, lo), (0.8 * e - 0.5, lo)),
                _rect((e - 0.5, e), (0.8 * e - 0.5, hi), (0.8 * e, hi), (e, e)),
                _rect((0.2 * e, lo - 0.5), (0.8 * e, lo - 0.5), (0.8 * e, lo), (0.2 * e, lo)),
                _rect((0.2 * e, hi + 0.5), (0.2 * e, hi), (0.8 * e, hi), (0.8 * e, hi + 0.5))]
    if scenario in ("crowd", SCN_CROWD):
        e = crowd_envsize(n_agents, radius)
        return [_rect((0.0, 0.0), (0.0, e), (e, e), (e, 0.0))]          # ALAN:291-292
    if scenario in ("circle", SCN_CIRCLE):
        e = circle_envsize(n_agents, radius)
        return [_rect((0.0, 0.0), (0.0, e), (e, e), (e, 0.0))]          # ALAN:327-328
    if scenario in ("doorway", SCN_DOORWAY):
        e = 10                                                            # env.py:41
        return [_rect((-15.0, 0.0), (-15.0, e), (e, e), (e, 0.0)),       # env.py:118
                _rect((2.0, 0.0), (2.5, 0.0), (2.5, 4.4), (2.0, 4.4)),   # env.py:121
                _rect((2.0, 5.6), (2.5, 5.6), (2.5, 10.0), (2.0, 10.0))]  # env.py:122
    raise ValueError("unknown scenario %r" % (scenario,))


# ---- counter-based draws on the host (the library's stream: Philox4x32-10 keyed by (seed, global arena id)) ----
RNG_BLOCKS = 6   # purpose word of the block-position draws (csrc/ca_math.h: RNG_POS .. RNG_ALAN are 0..5)


def _philox4x32(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on numpy uint64 arrays holding 32-bit words (same rounds and constants as csrc/ca_math.h)."""
    import numpy as np
    m32 = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3, k0, k1 = (np.asarray(v, np.uint64) & m32 for v in (c0, c1, c2, c3, k0, k1))
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        h0, l0, h1, l1 = p0 >> np.uint64(32), p0 & m32, p1 >> np.uint64(32), p1 & m32
        c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return c0, c1, c2, c3


def rng2(seed, arena, agent, purpose, seq=0):
    """Two uniforms in [0, 1) with 53 random bits each: the library's rng2 (csrc/ca_math.h), vectorised over
    `arena` (global arena ids)."""
    import numpy as np
    g = np.asarray(arena, np.int64).astype(np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    k1 = (np.uint64(seed >> 32) + (g >> np.uint64(32))) & np.uint64(0xFFFFFFFF)
    w = _philox4x32(g & np.uint64(0xFFFFFFFF), np.uint64(agent), np.uint64(purpose), np.uint64(seq),
                    np.uint64(seed & 0xFFFFFFFF), k1)
    k = 1.0 / 9007199254740992.0
    u0 = (((w[0] >> np.uint64(5)) << np.uint64(26)) | (w[1] >> np.uint64(6))).astype(np.float64) * k
    u1 = (((w[2] >> np.uint64(5)) << np.uint64(26)) | (w[3] >> np.uint64(6))).astype(np.float64) * k
    return u0, u1


def blocks_worlds(n_arenas, n_agents, radius=0.5, seed=0, arena_offset=0):
    """The "blocks" world of every arena: the border and FOUR RANDOM BLOCKS PER ARENA, as every simulator of the
    reference draws its own (ALAN:359-372; reset() redraws them, ALAN:92-100).  Block k of global arena g sits
    at (uniform(b, E - b), uniform(0, E)) with the uniforms rng2(seed, g, k, RNG_BLOCKS): keyed by the GLOBAL
    arena id, so the worlds do not depend on how arenas are sharded over GPUs."""
    import numpy as np
    e = envsize("blocks", n_agents, radius)
    b = e / (4 * 2)
    g = np.arange(n_arenas, dtype=np.int64) + int(arena_offset)
    draws = [rng2(seed, g, k, RNG_BLOCKS) for k in range(4)]
    worlds = []
    for a in range(n_arenas):
        polys = [_rect((0.0, 0.0), (0.0, e), (e, e), (e, 0.0))]
        for k in range(4):
            x = b + ((e - b) - b) * float(draws[k][0][a])
            y = 0.0 + (e - 0.0) * float(draws[k][1][a])
            polys.append(_rect((x - b / 2, y - b / 2), (x + b / 2, y - b / 2), (x + b / 2, y + b / 2), (x - b / 2, y + b / 2)))
        worlds.append(polys)
    return worlds


def obstacle_worlds(scenario, n_arenas, n_agents, radius=0.5, seed=0, arena_offset=0):
    """None when every arena of the scenario has the same obstacles (use obstacles()), else the list of per-arena
    polygon lists (only "blocks" draws its world at random)."""
    scenario = SCENARIO_NAMES.get(scenario, scenario)
    if scenario == "blocks":
        return blocks_worlds(n_arenas, n_agents, radius, seed, arena_offset)
    return None


def env_params():
    """The reference env's constants (env.py:27-44, 130, 359, 396, 478)."""
    return dict(time_step=1 / 60., neighbor_dist=1.5, max_neighbors=5, time_horizon=1.5,
                time_horizon_obst=1.5, radius=0.5, max_speed=1.0, max_step=1000,
                done_mode=DONE_XLESS, done_x_thresh=2.0, reward_scale=0.3,
                spawn_x0=5.0, spawn_x1=10.0, spawn_y0=0.0, spawn_y1=10.0,
                goal_x0=0.0, goal_x1=10.0, goal_y0=0.0, goal_y1=10.0)


def alan_params(n_agents, scenario="crowd"):
    """The ALAN simulator's constants (ALAN:15-20, 47, 59)."""
    e = envsize(scenario, n_agents)
    return dict(time_step=1 / 60., neighbor_dist=5.0, max_neighbors=10, time_horizon=1.5,
                time_horizon_obst=1.5, radius=0.5, max_speed=1.0,
                max_step=int((10 / (1 / 60.)) * n_agents), done_mode=DONE_GOAL, done_x_thresh=2.0,
                reward_scale=0.6, spawn_x0=0.0, spawn_x1=e, spawn_y0=0.0, spawn_y1=e,
                goal_x0=0.0, goal_x1=e, goal_y0=0.0, goal_y1=e)


def bench_params(n_agents, neighbor_dist, max_neighbors):
    """Synthetic random start/goal workload of SURVEY.md section 8d: the ALAN crowd recipe with a
    new goal drawn whenever one is reached, no episode cap."""
    p = alan_params(n_agents, "crowd")
    p.update(neighbor_dist=neighbor_dist, max_neighbors=max_neighbors, done_mode=DONE_REGOAL,
             max_step=0, reward_scale=0.3)
    return p


BENCH_CONFIGS = {
    # BASELINE.json configs[1..4]
    "C2": dict(n_arenas=1024, n_agents=16, neighbor_dist=1.5, max_neighbors=5),
    "C3": dict(n_arenas=4096, n_agents=64, neighbor_dist=5.0, max_neighbors=10),
    "C5": dict(n_arenas=256, n_agents=512, neighbor_dist=5.0, max_neighbors=10),
    # ALAN online learning (bench.py --mode alan; ALAN_true.py:106-123, 569-628): the paper's simulator constants
    # (neighborDist 5, maxNeighbors 10) on three batch shapes
    "A16": dict(n_arenas=1024, n_agents=16, neighbor_dist=5.0, max_neighbors=10),
    "A50": dict(n_arenas=1024, n_agents=50, neighbor_dist=5.0, max_neighbors=10),
    "A100": dict(n_arenas=1024, n_agents=100, neighbor_dist=5.0, max_neighbors=10, scenario="circle"),
    # mid-size arenas (one arena = a workgroup of two or three waves, brute-force neighbour scan): diagnostics
    "M128": dict(n_arenas=1024, n_agents=128, neighbor_dist=5.0, max_neighbors=10),
    "M180": dict(n_arenas=512, n_agents=180, neighbor_dist=5.0, max_neighbors=10),
}
