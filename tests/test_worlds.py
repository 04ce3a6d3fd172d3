"""A world per arena: the reference builds one simulator per environment and
draws the four blocks of the "blocks" scenario anew for each (ALAN_true.py:359-372), reset() redraws them
(:92-100) and the trainer averages over such worlds (Train_ALAN_action_space.py:53-66).

CPU part: the host-side Philox equals the oracle's; the oracle with one table per arena replays the three worlds of
tests/golden/alan_blocks.npz (recorded from the reference's own _init_world_blocks / run_sim(0)) AS ONE BATCH.
GPU part (marked): the HIP path does the same and matches the oracle on generated and hand-made per-arena worlds."""
import os

import numpy as np
import pytest

from tools import alan_actions

from collision_avoidance_amd import scenarios
from oracle import oracle as o
from tests import helpers as H


def load_worlds(golden_dir):
    g = np.load(os.path.join(golden_dir, "alan_blocks.npz"))
    W = int(g["n_worlds"])
    return [{k[3:]: g[k] for k in g.files if k.startswith("w%d_" % w)} for w in range(W)], int(g["n_agents"])


def test_host_philox_equals_oracle():
    for g, i, p, s, seed in [(0, 0, 6, 0, 0), (5, 3, 6, 0, 12345678901234), (2 ** 33 + 7, 1, 6, 2, 99), (4095, 63, 0, 1, 2 ** 63 + 5)]:
        u = scenarios.rng2(seed, np.array([g]), i, p, s)
        w = o.philox4x32((g & 0xffffffff, i, p, s), (seed & 0xffffffff, ((seed >> 32) + (g >> 32)) & 0xffffffff))
        assert u[0][0] == (((w[0] >> 5) << 26) | (w[1] >> 6)) * 2.0 ** -53
        assert u[1][0] == (((w[2] >> 5) << 26) | (w[3] >> 6)) * 2.0 ** -53


def test_blocks_worlds_are_keyed_by_global_arena():
    e = scenarios.envsize("blocks", 12)
    whole = scenarios.blocks_worlds(6, 12, seed=9)
    parts = scenarios.blocks_worlds(2, 12, seed=9, arena_offset=0) + scenarios.blocks_worlds(4, 12, seed=9, arena_offset=2)
    assert whole == parts                                                  # sharding does not change the worlds
    assert len({tuple(w[1][0]) for w in whole}) == 6                       # every arena its own blocks
    b = e / 8
    for w in whole:
        assert len(w) == 5 and w[0] == [(0.0, 0.0), (0.0, e), (e, e), (e, 0.0)]          # ALAN:359-360
        for blk in w[1:]:
            cx, cy = blk[0][0] + b / 2, blk[0][1] + b / 2
            assert b <= cx <= e - b and 0 <= cy <= e                        # ALAN:366-367
            assert abs(blk[2][0] - blk[0][0] - b) < 1e-12 and abs(blk[2][1] - blk[0][1] - b) < 1e-12


def batch_of_golden_worlds(make, worlds, n, set_field, F):
    """One batch whose arena w is golden world w: per-arena obstacle polygons, initial state of each world."""
    p = scenarios.alan_params(n, "blocks")
    env = make(len(worlds), n, p, [[q.tolist() for q in w["obst"]] for w in worlds])
    stack = lambda k, c: np.stack([w[k][:, c] for w in worlds])
    for f, k, c in ((F.FLD_POS_X, "pos0", 0), (F.FLD_POS_Y, "pos0", 1), (F.FLD_VEL_X, "vel0", 0), (F.FLD_VEL_Y, "vel0", 1),
                    (F.FLD_GOAL_X, "goal0", 0), (F.FLD_GOAL_Y, "goal0", 1), (F.FLD_GOAL2_X, "goal20", 0),
                    (F.FLD_GOAL2_Y, "goal20", 1), (F.FLD_PREF_X, "pref0", 0), (F.FLD_PREF_Y, "pref0", 1)):
        set_field(env, f, stack(k, c))
    return env, p


def replay_worlds(env, worlds, step, get, F):
    steps = max(int(w["steps"]) for w in worlds)
    for s in range(steps):
        step(env)
        if s % 5 == 0:
            px, py, vx, dn = get(env, F.FLD_POS_X), get(env, F.FLD_POS_Y), get(env, F.FLD_VEL_X), get(env, F.FLD_AGENT_DONE)
            for a, w in enumerate(worlds):
                if s < int(w["steps"]):
                    np.testing.assert_array_equal(px[a], w["pos"][s // 5][:, 0], err_msg="world %d step %d" % (a, s))
                    np.testing.assert_array_equal(py[a], w["pos"][s // 5][:, 1])
                    np.testing.assert_array_equal(vx[a], w["vel"][s // 5][:, 0])
                    np.testing.assert_array_equal(dn[a], w["done"][s // 5])
    for a, w in enumerate(worlds):
        if int(w["steps"]) == steps:
            np.testing.assert_array_equal(get(env, F.FLD_POS_X)[a], w["pos_last"][:, 0])
            np.testing.assert_array_equal(get(env, F.FLD_AGENT_DONE)[a], w["done_last"])


def test_oracle_replays_reference_block_worlds_as_one_batch(golden_dir):
    worlds, n = load_worlds(golden_dir)
    assert len({tuple(w["obst"][1].ravel()) for w in worlds}) == len(worlds)     # the reference drew different blocks

    def make(A, n, p, polys):
        e = o.OracleEnv(o.make_config(n_arenas=A, n_agents=n, max_obst_neighbors=16, **p))
        e.set_obstacles_per_arena(polys)
        e.init_scenario(o.SCN_BLOCKS)
        return e
    env, p = batch_of_golden_worlds(make, worlds, n, lambda e, f, v: e.set(f, v), o)
    for a, w in enumerate(worlds):   # processObstacles leaves these worlds uncut: the table is the 20 input vertices
        t = env.obstacle_table(arena=a)
        np.testing.assert_array_equal(np.stack([t["px"], t["py"]], -1)[:20], w["obst"].reshape(20, 2))
    replay_worlds(env, worlds, lambda e: e.orca_step(flags=o.F_FREEZE), lambda e, f: e.get(f), o)


def test_oracle_per_arena_tables_equal_single_arena_runs():
    """Arena a of a batch of generated worlds == the same world run alone (the tables do not leak into each other)."""
    A, N = 5, 12
    p = H.scenario_params("blocks", N)
    batch = H.make_oracle(A, N, "blocks", p, seed=3)
    batch.rollout(300, flags=o.F_STATS)
    for a in range(A):
        one = H.make_oracle(1, N, "blocks", p, seed=3, arena_offset=a)
        one.rollout(300, flags=o.F_STATS)
        for f in (o.FLD_POS_X, o.FLD_POS_Y, o.FLD_VEL_X, o.FLD_OBST_COUNT, o.FLD_OBST_IDX):
            np.testing.assert_array_equal(batch.get(f)[a], one.get(f)[0])
        ta, t1 = batch.obstacle_table(arena=a), one.obstacle_table()
        for k in ta:
            np.testing.assert_array_equal(ta[k], t1[k])
    assert len({tuple(batch.obstacle_table(arena=a)["px"][4:8]) for a in range(A)}) == A


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_per_arena_obstacle_golden_worlds_on_gpu(golden_dir):
    from collision_avoidance_amd import _lib
    worlds, n = load_worlds(golden_dir)

    def make(A, n, p, polys):
        return H.make_gpu(A, n, "blocks", p, max_obst_neighbors=16, polys=dict(per_arena=polys))
    env, p = batch_of_golden_worlds(make, worlds, n, lambda e, f, v: e.set(f, v), _lib)
    for a, w in enumerate(worlds):
        np.testing.assert_array_equal(env.obstacle_table(arena=a)["verts"][:20], w["obst"].reshape(20, 2))
    replay_worlds(env, worlds, lambda e: e.orca_step(freeze=True), lambda e, f: e.get(f), _lib)
    env.close()


def _squares(k, x0, y0, pitch, size):
    return [[(x0 + c * pitch, y0 + r * pitch), (x0 + c * pitch + size, y0 + r * pitch),
             (x0 + c * pitch + size, y0 + r * pitch + size), (x0 + c * pitch, y0 + r * pitch + size)]
            for r in range(k) for c in range(k)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,A,N,over,steps", [
    ("blocks12", 6, 12, dict(), 400),                       # generated worlds (one per arena), ALAN parameters
    ("blocks16x4", 9, 16, dict(), 300),                     # four arenas per wave, each with its own table
    ("blocks3", 7, 3, dict(), 200),                         # sixteen arenas per wave, ragged last wave
    ("blocks200", 2, 200, dict(), 60),                      # one arena per workgroup (grid scan), 12 edges... per arena
])
def test_per_arena_obstacle_worlds_match_oracle(name, A, N, over, steps):
    from collision_avoidance_amd import _lib
    p = H.scenario_params("blocks", N, **over)
    g = H.make_gpu(A, N, "blocks", p, seed=5)
    e = H.make_oracle(A, N, "blocks", p, seed=5)
    for a in range(A):
        tg, te = g.obstacle_table(arena=a), e.obstacle_table(arena=a)
        np.testing.assert_array_equal(tg["verts"], np.stack([te["px"], te["py"]], -1))
        np.testing.assert_array_equal(tg["next"], te["next"]); np.testing.assert_array_equal(tg["convex"], te["convex"])
    rng = np.random.RandomState(2)
    for s in range(steps):
        if s % 3 == 2:
            act = rng.uniform(-1, 1, (A, N)).astype(np.float32)
            g.step(act, stats=True); e.step(act, flags=o.F_OBS | o.F_STATS)
        else:
            g.orca_step(with_obs=True, stats=True); e.orca_step(flags=o.F_OBS | o.F_STATS)
        if s % 50 == 49 or s < 3:
            H.assert_state_equal(g, e, "%s step %d" % (name, s), obs=True)
    H.assert_state_equal(g, e, name + " end", obs=True)
    H.assert_stats_equal(g, e, name)
    assert g.stats()["obst_collisions"] >= 0 and (g.get(_lib.FLD_OBST_COUNT) > 0).any()
    g.close()


@pytest.mark.gpu
def test_per_arena_obstacle_ragged_and_wide_tables():
    """Worlds of different sizes in one batch: no obstacle at all, the plain border, a doorway-like world whose edges
    processObstacles cuts, and a grid of 81 small squares (324 edges: ids beyond a byte); then a common table again
    through ca_set_obstacles."""
    from collision_avoidance_amd import _lib
    N = 10
    p = H.scenario_params("crowd", N)
    E = scenarios.envsize("crowd", N)
    border = [(0.0, 0.0), (0.0, E), (E, E), (E, 0.0)]
    worlds = [[], [border],
              [border, [(2.0, 0.0), (2.5, 0.0), (2.5, 2.4), (2.0, 2.4)], [(2.0, 3.6), (2.5, 3.6), (2.5, E), (2.0, E)]],
              [border] + _squares(9, 0.3, 0.3, 0.65, 0.2)]
    # (a grid of squares puts up to 59 edges within range of an agent: more than any list holds -- accepted here on purpose, the
    # oracle truncates the same way; without the opt-in the step calls fail loudly: tests/test_gpu_overflow.py)
    g = H.make_gpu(4, N, "crowd", p, seed=8, polys=dict(per_arena=worlds), max_obst_neighbors=16, allow_obst_overflow=True)
    e = H.make_oracle(4, N, "crowd", p, seed=8, polys=dict(per_arena=worlds), max_obst_neighbors=16)
    assert [g.obstacle_table(arena=a)["verts"].shape[0] for a in range(4)][:2] == [0, 4]
    assert g.obstacle_table(arena=3)["verts"].shape[0] >= 4 + 81 * 4
    for a in range(4):
        np.testing.assert_array_equal(g.obstacle_table(arena=a)["next"], e.obstacle_table(cap=1024, arena=a)["next"])
    for s in range(250):
        g.orca_step(with_obs=True, stats=True); e.orca_step(flags=o.F_OBS | o.F_STATS)
        if s % 50 == 49:
            H.assert_state_equal(g, e, "ragged worlds step %d" % s, obs=True)
    H.assert_stats_equal(g, e, "ragged worlds")
    assert g.get(_lib.FLD_OBST_IDX).max() > 255          # ids beyond a byte were stored and read back
    g.set_obstacles([border]); e.set_obstacles([border])  # back to one table for every arena
    for s in range(60):
        g.orca_step(with_obs=True, stats=True); e.orca_step(flags=o.F_OBS | o.F_STATS)
    H.assert_state_equal(g, e, "common table again", obs=True)
    g.close()


@pytest.mark.gpu
def test_per_arena_obstacle_trainer_evaluation_over_random_block_worlds():
    """MCMC_trainer.evaluate_action (Train_ALAN_action_space.py:53-66): `num` rounds of reset() -- a NEW blocks world
    each (ALAN_true.py:92-100, 359-372) -- and run_sim(); here the rounds are the arenas of one handle, each with its
    own four random blocks, and equal the same rounds run one after the other."""
    from collision_avoidance_amd import alan
    acts = [(1, 0), (0.6, -0.8), (-0.5, 0.86)]
    mean_tt, ok = alan_actions.evaluate_actions(acts, numAgents=8, scenario="blocks", num=3, seed=11)
    seq = alan.Collision_Avoidance_Sim(numAgents=8, scenario="blocks", online_actions=acts, seed=11)
    tts, tabs = [], []
    for r in range(3):
        if r:
            seq.reset(acts)
        tabs.append(seq.vec.obstacle_table()["verts"].copy())
        tts.append(seq.run_sim(1)[2])
    assert abs(mean_tt - float(np.mean(tts))) < 1e-12 and 0 <= ok <= 3
    assert not np.array_equal(tabs[0], tabs[1]) and not np.array_equal(tabs[1], tabs[2])      # reset() drew new blocks
    batch = alan.Collision_Avoidance_Sim(numAgents=8, scenario="blocks", online_actions=acts, seed=11, n_arenas=3)
    for r in range(3):
        np.testing.assert_array_equal(batch.vec.obstacle_table(arena=r)["verts"], tabs[r])
    batch.vec.close(); seq.vec.close()


def test_separated_start_crowd_is_the_crowd_with_redrawn_overlaps():
    """SURVEY 8d bench variant: starts at least 2 r apart by redrawing (sequence 1, 2, ...) -- agents that did not
    overlap an earlier agent keep the plain crowd's start, goals and headings are the plain crowd's."""
    from collision_avoidance_amd import _lib
    assert scenarios.SCENARIO_IDS["crowd_separated"] == o.SCN_CROWD_SEPARATED == _lib.SCN_CROWD_SEPARATED == 7
    A, N = 16, 64
    p = scenarios.bench_params(N, 5.0, 10)
    sep = H.make_oracle(A, N, "crowd_separated", p, seed=4)
    plain = H.make_oracle(A, N, "crowd", p, seed=4)
    x, y = sep.get(o.FLD_POS_X).astype(np.float64), sep.get(o.FLD_POS_Y).astype(np.float64)
    d = np.hypot(x[:, :, None] - x[:, None, :], y[:, :, None] - y[:, None, :]) + 10 * np.eye(N)[None]
    assert d.min() >= 1.0 - 1e-6                                           # 2 r = 1: nobody starts inside anybody
    same = (sep.get(o.FLD_POS_X) == plain.get(o.FLD_POS_X)) & (sep.get(o.FLD_POS_Y) == plain.get(o.FLD_POS_Y))
    assert 0.5 < same.mean() < 0.95 and same[:, 0].all()                   # agent 0 never moves; later ones sometimes
    for f in (o.FLD_GOAL_X, o.FLD_GOAL_Y, o.FLD_VEL_X, o.FLD_VEL_Y):
        np.testing.assert_array_equal(sep.get(f), plain.get(f))
    px, py = plain.get(o.FLD_POS_X).astype(np.float64), plain.get(o.FLD_POS_Y).astype(np.float64)
    dp = np.hypot(px[:, :, None] - px[:, None, :], py[:, :, None] - py[:, None, :]) + 10 * np.eye(N)[None]
    assert dp.min() < 1.0                                                  # the plain crowd does start with overlaps


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,n_edges", [(5, 6, 12), (10, 12, 16), (5, 40, 14)])
def test_small_world_with_many_edges_in_range_on_the_register_line_kernel(K, N, n_edges):
    """Worlds of at most 16 edges run on the register-line solve kernel, which has four obstacle-line slots; an agent
    with MORE obstacle neighbours than that (RVO2 keeps every edge in range, env.py:249, 301-318) is solved apart, exactly
    (ca_step.h solve_many_obstacles).  Here nearly every agent is such an agent: a small round room (a clockwise n-gon of
    radius 2.2, so its walls face inwards and most of them are within range of everybody), forced onto the lane kernel."""
    import math
    import os
    from collision_avoidance_amd import _lib
    R = 2.2 if N <= 12 else 2.8
    room = [(5.0 + R * math.cos(-2 * math.pi * k / n_edges), 5.0 + R * math.sin(-2 * math.pi * k / n_edges)) for k in range(n_edges)]
    p = H.scenario_params("crowd", N, max_neighbors=K, neighbor_dist=3.0)
    lo, hi = 5.0 - 0.5 * R, 5.0 + 0.5 * R
    p.update(spawn_x0=lo, spawn_x1=hi, spawn_y0=lo, spawn_y1=hi, goal_x0=lo, goal_x1=hi, goal_y0=lo, goal_y1=hi, done_mode=2, max_step=0)
    old = os.environ.get("CA_QUAD")
    os.environ["CA_QUAD"] = "0"     # (a batch this small would otherwise take the four-lanes kernel, which has its own 16-entry variant)
    try:
        A = 24
        g = H.make_gpu(A, N, "crowd", p, seed=5, polys=[room])
        e = H.make_oracle(A, N, "crowd", p, seed=5, polys=[room])
    finally:
        if old is None:
            del os.environ["CA_QUAD"]
        else:
            os.environ["CA_QUAD"] = old
    assert g.launch_info()["lanes_per_agent"] == 1 and g.S == n_edges
    g.reset(); e.reset()                      # into the room
    rng = np.random.RandomState(2)
    many = 0
    for s in range(60):
        act = rng.uniform(-1.0, 1.0, (A, N)).astype(np.float32)
        g.step(act, stats=True); e.step(act, flags=o.F_OBS | o.F_STATS)
        many += int((e.get(o.FLD_OBST_COUNT) > 4).sum())
        if s % 20 == 19:
            H.assert_state_equal(g, e, "round room step %d" % s, obs=True, reward=True)
    H.assert_stats_equal(g, e, "round room")
    assert many > (0.3 if N <= 12 else 0.05) * 60 * A * N, many          # the stage was really exercised
    assert g.stats()["obst_overflow"] == 0
    g.close()


@pytest.mark.gpu
def test_large_arena_with_sixteen_obstacle_neighbours_is_accepted():
    """300 agents per arena with the default obstacle-list capacity of a 14-edge world: the LDS line table would need more
    than the 160 KB of a CU (round 3 refused the configuration with CA_ERANGE); the register-line kernel with the
    many-obstacles stage takes it."""
    N, A = 300, 2
    p = H.scenario_params("crowd", N, max_neighbors=5, neighbor_dist=2.0)
    walls = scenarios.obstacles("doorway", 10)
    p.update(spawn_x0=3.0, spawn_x1=9.5, spawn_y0=0.5, spawn_y1=9.5, goal_x0=-10.0, goal_x1=1.0, goal_y0=1.0, goal_y1=9.0)
    g = H.make_gpu(A, N, "crowd", p, seed=3, polys=walls)
    e = H.make_oracle(A, N, "crowd", p, seed=3, polys=walls)
    assert g.S == 12 and g.launch_info()["lanes_per_agent"] == 1      # (12 edges before processObstacles cuts two of them)
    g.reset(); e.reset()
    for s in range(40):
        g.orca_step(stats=True, with_obs=(s == 39)); e.orca_step(flags=o.F_STATS | (o.F_OBS if s == 39 else 0))
    H.assert_state_equal(g, e, "300 agents behind the doorway", obs=True)
    H.assert_stats_equal(g, e, "300 agents behind the doorway")
    g.close()


@pytest.mark.gpu
def test_large_batches_of_many_edge_worlds_take_the_register_lines():
    """The two-way tube of the ALAN runs ("deadlock", 42 edges + cuts, ALAN:418-455; every sixth agent-step has more than four
    edges in range): a batch the chip holds at once with the LDS line table keeps the table, a larger one takes the register
    lines with the many-obstacles stage (ca_env.hip pick_variant: five 64-lane table workgroups fit a CU) -- same bits either
    way: full steps with observation, then ALAN online steps inside the same launch, every arena against the oracle."""
    from collision_avoidance_amd import _lib, alan
    N = 50
    p = H.scenario_params("deadlock", N, max_step=400)
    small = H.make_gpu(1100, N, "deadlock", p, seed=3)       # (fewer than 1024 waves would take the four-lanes kernel)
    assert small.launch_info()["lanes_per_agent"] == 1
    assert small.launch_info()["lds_bytes"] == 64 * ((10 + 16) * 16 + 32)          # the line table: [K + S][lanes] float4 + staging
    small.close()
    A = 1536
    g = H.make_gpu(A, N, "deadlock", p, seed=3)
    e = H.make_oracle(A, N, "deadlock", p, seed=3)
    assert g.launch_info()["lanes_per_agent"] == 1 and g.launch_info()["lds_bytes"] == 2 * 14 * 16 * 16 + 64 * 32   # the wave's LP3 pool + staging
    rng = np.random.RandomState(4)
    many = 0
    for s in range(24):
        act = rng.uniform(-1.0, 1.0, (A, N)).astype(np.float32)
        g.step(act, stats=True)
        e.step_mt(act, flags=o.F_OBS | o.F_STATS, n_threads=8)
        many += int((e.get(o.FLD_OBST_COUNT) > 4).sum())
    H.assert_state_equal(g, e, "tube, full steps", obs=True, reward=True)
    assert many > 0.05 * 24 * A * N, many
    g.alan_configure(alan.DEFAULT_ACTIONS); e.alan_configure(alan.DEFAULT_ACTIONS)
    g.profile(1); g.profile_read()
    for s in range(6):
        g.alan_step(stats=True); e.alan_step(flags=o.F_STATS)
    prof = g.profile_read(); g.profile(0)
    assert prof["step_kernel"][0] == 6 and prof["reset_kernels"][0] == 0, prof     # the bandit ran inside the solve launch
    H.assert_state_equal(g, e, "tube, ALAN steps")
    assert np.array_equal(g.get(_lib.FLD_ALAN_WEIGHTS).view(np.uint64), e.get(o.FLD_ALAN_WEIGHTS).view(np.uint64)), "ALAN weights"
    H.assert_stats_equal(g, e, "tube")
    assert g.stats()["obst_overflow"] == 0
    g.close()


QUOTED = [  # the batches of the reference's own worlds whose rates DESIGN.md / BASELINE.md quote
    ("doorway", 16384, 10),      # the drop-in env's configuration (env.py:26-123): register lines, dense packing, dense observation
    ("congested", 4096, 50),     # ALAN:195-208: register lines, obstacle lists of 16
    ("deadlock", 4096, 50),      # ALAN:418-455: the same kernel by the residency rule, many-obstacle agents eight at a time
    ("blocks", 8192, 20),        # ALAN:359-372: a world per arena
]


@pytest.mark.gpu
@pytest.mark.parametrize("scenario,A,N", QUOTED, ids=[c[0] for c in QUOTED])
def test_reference_world_batches_at_the_quoted_sizes(scenario, A, N):
    """Every arena of the quoted batch shapes against the oracle, bit for bit: an ORCA-only rollout, then full steps with
    actions, reward and the observation, then ALAN online steps (the oracle steps the batch on a thread pool)."""
    import os
    from collision_avoidance_amd import _lib, alan
    nt = min(16, os.cpu_count() or 1)
    p = H.scenario_params(scenario, N)
    g = H.make_gpu(A, N, scenario, p, seed=6)
    e = H.make_oracle(A, N, scenario, p, seed=6)
    assert g.launch_info()["lanes_per_agent"] == 1 and g.launch_info()["lds_bytes"] < 16 * 1024     # a register-line kernel
    g.rollout(12, stats=True)
    e.rollout(12, flags=o.F_STATS, n_threads=nt)
    H.assert_state_equal(g, e, scenario + " batch, ORCA rollout")
    rng = np.random.RandomState(8)
    for s in range(3):
        act = rng.uniform(-1.0, 1.0, (A, N)).astype(np.float32)
        g.step(act, stats=True)
        e.step_mt(act, flags=o.F_OBS | o.F_STATS, n_threads=nt)
    H.assert_state_equal(g, e, scenario + " batch, full steps", obs=True, reward=True)
    if scenario != "doorway":   # (the env's own world has no ALAN run)
        g.alan_configure(alan.DEFAULT_ACTIONS); e.alan_configure(alan.DEFAULT_ACTIONS)
        for s in range(3):
            g.alan_step(stats=True); e.alan_step(flags=o.F_STATS)
        H.assert_state_equal(g, e, scenario + " batch, ALAN steps")
    H.assert_stats_equal(g, e, scenario + " batch")
    assert g.stats()["obst_overflow"] == 0
    g.close()
