"""The oracle's ALAN online step against runs of the reference's own online_step
(tests/golden/alan_online.npz): action draws, trajectories, bandit weights, arrival times, TTime."""
import os

import numpy as np
import pytest

from collision_avoidance_amd import scenarios
from oracle import oracle as o


# The reference forms an agent's preferred velocity as (cos(a), sin(a)), a = np.arctan2(dy, dx) in fp64 (ALAN_true.py:489-495,
# env.py:156-162): `a` is quantised to 2^-51 near +-pi, so a component that should be 1.7e-8 comes out with an ABSOLUTE
# error of up to ~4.4e-16 -- and WHICH error depends on the machine: numpy's arctan2 is SIMD-dispatched (AVX512 SVML in the
# container that wrote the fixtures: it differs from libm's atan2 in the last bit in 7.7 % of 200 000 random directions).
# The build takes the direction as the normalised vector (accurate to an fp64 ulp of the component itself).  The two
# agree bit for bit after rounding to fp32 unless a component is tiny (|c| < ~1e-7: an agent heading exactly along an
# axis: the "incoming" and "deadlock" worlds).  So: positions exact; velocity components exact, except that a component
# below 1e-6 in magnitude may differ by the quantisation plus the one fp32 ulp it can push the rounding over.
PREF_ANGLE_TOL = 4.5e-16
TINY_COMPONENT = 1e-6


def assert_vel_close(actual, golden, msg=""):
    """Exact, or -- see PREF_ANGLE_TOL -- within the reference's own angle quantisation; returns 1 for a non-exact match."""
    if np.array_equal(actual, golden):
        return 0
    a, g = actual.astype(np.float64), golden.astype(np.float64)
    bad = a != g
    assert (np.abs(g[bad]) < TINY_COMPONENT).all(), (msg, a[bad], g[bad])
    assert (np.abs(a[bad] - g[bad]) <= PREF_ANGLE_TOL + 2.0 ** -23 * np.abs(g[bad])).all(), (msg, a[bad], g[bad])
    return 1


def load_case(golden_dir, ci):
    g = np.load(os.path.join(golden_dir, "alan_online.npz"))
    key = "c%d_" % ci
    return {k[len(key):]: g[k] for k in g.files if k.startswith(key)}


def setup_env(make_env, c, set_field, F):
    """Common initial state for oracle and GPU replays."""
    scen = str(c["scenario"])
    n = c["pos0"].shape[0]
    p = scenarios.alan_params(n, scen)
    env = make_env(n, scen, p)
    for f, v in ((F.FLD_POS_X, c["pos0"][:, 0]), (F.FLD_POS_Y, c["pos0"][:, 1]), (F.FLD_VEL_X, c["vel0"][:, 0]),
                 (F.FLD_VEL_Y, c["vel0"][:, 1]), (F.FLD_GOAL_X, c["goal0"][:, 0]), (F.FLD_GOAL_Y, c["goal0"][:, 1]),
                 (F.FLD_GOAL2_X, c["goal20"][:, 0]), (F.FLD_GOAL2_Y, c["goal20"][:, 1])):
        set_field(env, f, v)
    return env, n, p


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_online_step_matches_reference(golden_dir, ci):
    c = load_case(golden_dir, ci)

    def make(n, scen, p):
        e = o.OracleEnv(o.make_config(n_arenas=1, n_agents=n, max_obst_neighbors=16, **p))
        e.set_obstacles(scenarios.obstacles(scen, n))
        e.init_scenario(scenarios.SCENARIO_IDS[scen])
        return e
    env, n, p = setup_env(make, c, lambda e, f, v: e.set(f, v), o)
    env.alan_configure(c["actions"])
    assert p["max_step"] == int(c["max_step"])
    steps = c["u"].shape[0]
    for s in range(steps):
        env.alan_step(c["u"][s], prec=o.PREC_F64)
        np.testing.assert_array_equal(env.get(o.FLD_POS_X)[0], c["pos"][s][:, 0], err_msg="step %d" % s)
        np.testing.assert_array_equal(env.get(o.FLD_POS_Y)[0], c["pos"][s][:, 1])
        np.testing.assert_array_equal(env.get(o.FLD_VEL_X)[0], c["vel"][s][:, 0])
        assert_vel_close(env.get(o.FLD_VEL_Y)[0], c["vel"][s][:, 1], "step %d" % s)
        np.testing.assert_array_equal(env.get(o.FLD_AGENT_DONE)[0], c["done"][s])
        if s % 10 == 0:
            np.testing.assert_allclose(env.get(o.FLD_ALAN_WEIGHTS)[0], c["w"][s // 10], rtol=0, atol=1e-13)
            np.testing.assert_array_equal(env.get(o.FLD_ALAN_TIMES)[0], c["t"][s // 10])   # same fp64 additions
    np.testing.assert_allclose(env.get(o.FLD_ALAN_WEIGHTS)[0], c["w_last"], rtol=0, atol=1e-13)
    # arrival times and TTime (ALAN_true.py:125-131, 559)
    from collision_avoidance_amd import alan
    arrive = env.get(o.FLD_ARRIVE_STEP)[0]
    times = alan.agents_time(arrive, env.get(o.FLD_AGENT_DONE)[0], p["time_step"], p["max_step"])
    np.testing.assert_allclose(times, c["agents_time"], rtol=0, atol=1e-12)
    assert abs(alan.ttime(times) - float(c["TTime"])) < 1e-12
    assert abs(alan.min_ttime(c["pos0"], c["goal0"], p["max_speed"]) - float(c["min_TTime"])) < 1e-12


def test_softmax_draw_pieces():
    for x in np.linspace(-6, 6, 2001):
        assert abs(o.exp64(x) - np.exp(x)) <= 3e-16 * np.exp(x)
    # counter-based stream when no uniforms are injected: deterministic and shard-invariant
    n = 8
    p = scenarios.alan_params(n, "crowd")
    acts = [(1, 0), (0, 1), (-1, 0), (0, -1)]

    def run(A, off):
        e = o.OracleEnv(o.make_config(n_arenas=A, n_agents=n, seed=5, arena_offset=off, **p))
        e.set_obstacles(scenarios.obstacles("crowd", n)); e.init_scenario(o.SCN_CROWD); e.alan_configure(acts)
        for _ in range(50):
            e.alan_step()
        return e.get(o.FLD_POS_X), e.get(o.FLD_ALAN_ACTION)
    px, act = run(3, 0)
    px2, act2 = run(1, 2)
    np.testing.assert_array_equal(px[2], px2[0]); np.testing.assert_array_equal(act[2], act2[0])
    assert len(np.unique(act)) > 1


def test_freeze_is_the_break_of_run_sim():
    """ORC_F_FREEZE: an arena stepped inside a batch until everything is done ends in the state it has when
    it is run alone and the loop breaks at its own end (ALAN_true.py:118-123)."""
    n = 6
    p = scenarios.alan_params(n, "crowd")
    p["max_step"] = 900

    def make(A, off):
        e = o.OracleEnv(o.make_config(n_arenas=A, n_agents=n, seed=11, arena_offset=off, **p))
        e.set_obstacles(scenarios.obstacles("crowd", n)); e.init_scenario(o.SCN_CROWD)
        e.alan_configure([(1, 0), (0, 1), (0, -1)])
        return e
    batch = make(4, 0)
    for _ in range(900):
        batch.alan_step(flags=o.F_FREEZE | o.F_STATS)
    assert batch.get(o.FLD_ARENA_DONE).all()
    total = 0
    for a in range(4):
        single = make(1, a)
        while not single.get(o.FLD_ARENA_DONE)[0]:
            single.alan_step(flags=o.F_STATS)
        for f in (o.FLD_POS_X, o.FLD_VEL_Y, o.FLD_ALAN_WEIGHTS, o.FLD_ALAN_TIMES, o.FLD_ARRIVE_STEP, o.FLD_STEP_COUNT):
            np.testing.assert_array_equal(batch.get(f)[a], single.get(f)[0])
        total += single.stats()["agent_steps"]
    assert batch.stats()["agent_steps"] == total
    assert len(set(batch.get(o.FLD_STEP_COUNT).tolist())) > 1


def load_orca_case(golden_dir, ci):
    g = np.load(os.path.join(golden_dir, "alan_orca.npz"))
    key = "c%d_" % ci
    return {k[len(key):]: g[k] for k in g.files if k.startswith(key)}


def replay_orca_episode(env, c, F, step, get):
    """run_sim(mode=0) of the reference (ALAN_true.py:106-123, 631-636) replayed: positions and velocities every
    fifth step, done flags, arrival times, TTime."""
    from collision_avoidance_amd import alan
    steps = int(c["steps"])
    for s in range(steps):
        step(env)
        if s % 5 == 0:
            np.testing.assert_array_equal(get(env, F.FLD_POS_X)[0], c["pos"][s // 5][:, 0], err_msg="step %d" % s)
            np.testing.assert_array_equal(get(env, F.FLD_POS_Y)[0], c["pos"][s // 5][:, 1])
            np.testing.assert_array_equal(get(env, F.FLD_VEL_X)[0], c["vel"][s // 5][:, 0])
            assert_vel_close(get(env, F.FLD_VEL_Y)[0], c["vel"][s // 5][:, 1], "step %d" % s)
            np.testing.assert_array_equal(get(env, F.FLD_AGENT_DONE)[0], c["done"][s // 5])
    np.testing.assert_array_equal(get(env, F.FLD_POS_X)[0], c["pos_last"][:, 0])
    np.testing.assert_array_equal(get(env, F.FLD_AGENT_DONE)[0], c["done_last"])
    assert int(get(env, F.FLD_STEP_COUNT)[0]) == steps and bool(get(env, F.FLD_ARENA_DONE)[0]) == bool(c["success"])
    times = alan.agents_time(get(env, F.FLD_ARRIVE_STEP)[0], get(env, F.FLD_AGENT_DONE)[0], 1 / 60., int(c["max_step"]))
    np.testing.assert_allclose(times, c["agents_time"], rtol=0, atol=1e-12)
    assert abs(alan.ttime(times) - float(c["TTime"])) < 1e-9


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_plain_orca_episode_matches_reference(golden_dir, ci):
    c = load_orca_case(golden_dir, ci)

    def make(n, scen, p):
        e = o.OracleEnv(o.make_config(n_arenas=1, n_agents=n, max_obst_neighbors=16, **p))
        e.set_obstacles(scenarios.obstacles(scen, n))
        e.init_scenario(scenarios.SCENARIO_IDS[scen])
        return e
    env, n, p = setup_env(make, c, lambda e, f, v: e.set(f, v), o)
    env.set(o.FLD_PREF_X, c["pref0"][:, 0]); env.set(o.FLD_PREF_Y, c["pref0"][:, 1])   # set by _init_world (update_pref_vel)
    replay_orca_episode(env, c, o, lambda e: e.orca_step(flags=0), lambda e, f: e.get(f))


# ---- episodes that END (round 5): run_sim itself, recorded from inside its own loop -------------------------------
N_FINISHED = 7


def load_finished_case(golden_dir, ci):
    g = np.load(os.path.join(golden_dir, "alan_finished.npz"))
    key = "c%d_" % ci
    return {k[len(key):]: g[k] for k in g.files if k.startswith(key)}


def replay_finished_episode(env, c, F, step, get, p):
    """The reference's run_sim(mode) (ALAN_true.py:106-131) replayed to its END: `step(env, s)` advances one step of
    the loop (orca_step or online_step + counter + done_test); the arena's done flag must come true at exactly the
    step where the reference's loop broke (success) or never (the run into max_step); then total_time, the arrival
    times, TTime, min_TTime and the swapped targets of :547-566."""
    from collision_avoidance_amd import alan
    steps, success = int(c["steps"]), bool(c["success"])
    inexact = 0
    for s in range(steps):
        assert not bool(get(env, F.FLD_ARENA_DONE)[0]), "arena done before step %d of %d" % (s, steps)
        step(env, s)
        np.testing.assert_array_equal(get(env, F.FLD_AGENT_DONE)[0], c["done"][s], err_msg="step %d" % s)
        if s % 5 == 0:
            np.testing.assert_array_equal(get(env, F.FLD_POS_X)[0], c["pos"][s // 5][:, 0], err_msg="step %d" % s)
            np.testing.assert_array_equal(get(env, F.FLD_POS_Y)[0], c["pos"][s // 5][:, 1])
            inexact += assert_vel_close(get(env, F.FLD_VEL_X)[0], c["vel"][s // 5][:, 0], "step %d" % s)
            inexact += assert_vel_close(get(env, F.FLD_VEL_Y)[0], c["vel"][s // 5][:, 1], "step %d" % s)
    assert inexact == 0 or str(c["scenario"]) in ("incoming", "deadlock"), inexact   # only axis-aligned worlds meet the quantisation
    np.testing.assert_array_equal(get(env, F.FLD_POS_X)[0], c["pos_last"][:, 0])
    np.testing.assert_array_equal(get(env, F.FLD_POS_Y)[0], c["pos_last"][:, 1])
    assert_vel_close(get(env, F.FLD_VEL_X)[0], c["vel_last"][:, 0])
    assert_vel_close(get(env, F.FLD_VEL_Y)[0], c["vel_last"][:, 1])
    # success -> break (:121-123); otherwise the loop ran out at max_step (the build's arena_done then is the step cap)
    assert int(get(env, F.FLD_STEP_COUNT)[0]) == steps
    assert bool(get(env, F.FLD_AGENT_DONE)[0].all()) == success
    assert bool(get(env, F.FLD_ARENA_DONE)[0]) and (success or steps == int(c["max_step"]))
    assert steps * p["time_step"] == float(c["total_time"])                                   # :125
    times = alan.agents_time(get(env, F.FLD_ARRIVE_STEP)[0], get(env, F.FLD_AGENT_DONE)[0], p["time_step"], int(c["max_step"]))
    np.testing.assert_allclose(times, c["agents_time"], rtol=0, atol=1e-12)
    assert abs(alan.ttime(times) - float(c["TTime"])) < 1e-9                                  # :126-130
    assert abs(alan.min_ttime(c["pos0"], c["goal0"], p["max_speed"]) - float(c["min_TTime"])) < 1e-12
    np.testing.assert_array_equal(get(env, F.FLD_GOAL_X)[0], c["goal_last"][:, 0])            # :563-564
    np.testing.assert_array_equal(get(env, F.FLD_GOAL_Y)[0], c["goal_last"][:, 1])


def test_finished_fixture_reaches_the_end_of_run_sim(golden_dir):
    succ = [int(load_finished_case(golden_dir, ci)["success"]) for ci in range(N_FINISHED)]
    modes = [int(load_finished_case(golden_dir, ci)["mode"]) for ci in range(N_FINISHED)]
    assert succ == [1, 1, 1, 1, 0, 1, 1] and modes == [0, 0, 0, 0, 0, 1, 1]
    c = load_finished_case(golden_dir, 4)
    assert int(c["steps"]) == int(c["max_step"]) and c["done"][-1].min() == 0     # ran into the cap, one agent on its way


@pytest.mark.parametrize("ci", range(N_FINISHED))
def test_finished_episode_matches_reference(golden_dir, ci):
    c = load_finished_case(golden_dir, ci)

    def make(n, scen, p):
        e = o.OracleEnv(o.make_config(n_arenas=1, n_agents=n, max_obst_neighbors=16, **p))
        e.set_obstacles(scenarios.obstacles(scen, n))
        e.init_scenario(scenarios.SCENARIO_IDS[scen])
        return e
    env, n, p = setup_env(make, c, lambda e, f, v: e.set(f, v), o)
    env.set(o.FLD_PREF_X, c["pref0"][:, 0]); env.set(o.FLD_PREF_Y, c["pref0"][:, 1])
    if int(c["mode"]) == 1:
        env.alan_configure(c["actions"])
        replay_finished_episode(env, c, o, lambda e, s: e.alan_step(c["u"][s], prec=o.PREC_F64), lambda e, f: e.get(f), p)
        np.testing.assert_allclose(env.get(o.FLD_ALAN_WEIGHTS)[0], c["w_last"], rtol=0, atol=1e-13)
        np.testing.assert_array_equal(env.get(o.FLD_ALAN_TIMES)[0], c["t_last"])
    else:
        replay_finished_episode(env, c, o, lambda e, s: e.orca_step(flags=0), lambda e, f: e.get(f), p)
