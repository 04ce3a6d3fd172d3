"""GPU parity of the ALAN online step (ca_alan_step, reference ALAN_true.py:569-628) and of the
per-arena freeze: the HIP path against the oracle (bit-exact, fp64 weights included), against the
recorded runs of the reference (tests/golden/alan_online.npz), and the Collision_Avoidance_Sim drop-in."""
import numpy as np
import pytest

from tools import alan_actions

from collision_avoidance_amd import _lib, alan, scenarios
from oracle import oracle as o
from tests import helpers as H
from tests.test_oracle_alan import assert_vel_close, load_case, setup_env

pytestmark = pytest.mark.gpu

ACTS9 = [(1, 0), (0.70711, 0.70711), (0, 1), (-0.70711, 0.70711), (-1, 0), (-0.70711, -0.70711), (0, -1),
         (0.70711, -0.70711), (0.3, 0.1)]


def test_numerics_contract_exp64():
    env = H.make_gpu(1, 4, "crowd", H.scenario_params("crowd", 4))
    x = np.concatenate([np.linspace(-40, 40, 20001), np.random.RandomState(0).uniform(-700, 700, 20000), [0.0]])
    out = np.empty_like(x)
    env._call("ca_debug_math", env.h, 5, x.ctypes.data, out.ctypes.data, len(x))
    np.testing.assert_array_equal(out, np.array([o.exp64(v) for v in x]))
    env.close()


def _assert_alan_equal(g, e, what):
    H.assert_state_equal(g, e, what, reward=True)
    H._eq(g.get(_lib.FLD_ALAN_ACTION), e.get(o.FLD_ALAN_ACTION), what + " action")
    gw, ew = g.get(_lib.FLD_ALAN_WEIGHTS), e.get(o.FLD_ALAN_WEIGHTS)
    assert np.array_equal(gw.view(np.uint64), ew.view(np.uint64)), what + " weights"
    gt, et = g.get(_lib.FLD_ALAN_TIMES), e.get(o.FLD_ALAN_TIMES)
    assert np.array_equal(gt.view(np.uint64), et.view(np.uint64)), what + " times"
    H._eq(g.get(_lib.FLD_ARRIVE_STEP), e.get(o.FLD_ARRIVE_STEP), what + " arrive_step")


@pytest.mark.parametrize("scenario,A,N,acts,steps", [
    ("crowd", 24, 16, alan.DEFAULT_ACTIONS, 400),       # 8 actions: numpy's 8-accumulator sum
    ("circle", 6, 24, alan.DEFAULT_ACTIONS[:3], 300),   # < 8 actions: sequential sum
    ("deadlock", 4, 10, ACTS9, 300),                    # 9 actions: accumulators + tail
    ("crowd", 3, 200, [(1, 0), (0, 1)] * 16, 60),       # 32 actions, arenas larger than a workgroup of the ALAN kernels
])
def test_alan_step_bit_exact(scenario, A, N, acts, steps):
    p = H.scenario_params(scenario, N)
    g = H.make_gpu(A, N, scenario, p, seed=3)
    e = H.make_oracle(A, N, scenario, p, seed=3)
    g.alan_configure(acts); e.alan_configure(acts)
    for s in range(steps):
        with_obs = s % 50 == 49
        g.alan_step(with_obs=with_obs, stats=True)
        e.alan_step(flags=(o.F_OBS if with_obs else 0) | o.F_STATS)
        if s % 25 == 24 or s < 3:
            _assert_alan_equal(g, e, "%s step %d" % (scenario, s))
            if with_obs:
                H._eq(g.get(_lib.FLD_OBS), e.get(o.FLD_OBS), "obs step %d" % s)
    H.assert_stats_equal(g, e, scenario)
    assert len(np.unique(g.get(_lib.FLD_ALAN_ACTION))) > 1
    g.close()


def test_alan_injected_uniforms_and_device_pointer():
    import torch
    A, N = 5, 12
    p = H.scenario_params("crowd", N)
    g = H.make_gpu(A, N, "crowd", p, seed=1)
    g2 = H.make_gpu(A, N, "crowd", p, seed=1)
    e = H.make_oracle(A, N, "crowd", p, seed=1)
    for env in (g, g2, e):
        env.alan_configure(alan.DEFAULT_ACTIONS)
    rng = np.random.RandomState(4)
    for s in range(120):
        u = rng.random_sample((A, N))
        if s == 7:
            u[0, :] = 0.0
            u[1, :] = np.nextafter(1.0, 0.0)
        g.alan_step(u)
        g2.alan_step(torch.as_tensor(u, device="cuda"))
        e.alan_step(u)
    _assert_alan_equal(g, e, "host uniforms")
    _assert_alan_equal(g2, e, "device uniforms")
    g.close(); g2.close()


def test_freeze_stops_each_arena_at_its_own_end():
    """CA_F_FREEZE == the `break` of run_sim (ALAN:121-123) per arena, for both step kinds."""
    A, N = 12, 8
    p = H.scenario_params("crowd", N, max_step=1000)     # the cap ends (and freezes) whatever is still running
    for mode in ("alan", "orca"):
        g = H.make_gpu(A, N, "crowd", p, seed=9)
        e = H.make_oracle(A, N, "crowd", p, seed=9)
        g.alan_configure(alan.DEFAULT_ACTIONS); e.alan_configure(alan.DEFAULT_ACTIONS)
        seen_partial = False
        for s in range(1500):
            if mode == "alan":
                g.alan_step(freeze=True, stats=True, with_obs=(s % 100 == 0))
                e.alan_step(flags=o.F_FREEZE | o.F_STATS | (o.F_OBS if s % 100 == 0 else 0))
            else:
                g.orca_step(freeze=True, stats=True)
                e.orca_step(flags=o.F_FREEZE | o.F_STATS)
            if s % 100 == 99:
                d = e.get(o.FLD_ARENA_DONE)
                seen_partial |= bool(d.any() and not d.all())
                _assert_alan_equal(g, e, "%s freeze step %d" % (mode, s))
                if d.all():
                    break
        assert seen_partial and e.get(o.FLD_ARENA_DONE).all()
        steps = g.get(_lib.FLD_STEP_COUNT)
        assert len(np.unique(steps)) > 1                     # arenas ended at different steps ...
        arrive = g.get(_lib.FLD_ARRIVE_STEP)
        ok = g.get(_lib.FLD_AGENT_DONE).all(axis=1)
        np.testing.assert_array_equal(steps[ok], arrive.max(axis=1)[ok])   # ... each at its last arrival
        assert ok.sum() >= A // 2 and (steps[~ok] == 1000).all()             # ... or at the cap
        H.assert_stats_equal(g, e, mode + " freeze")
        g.close()


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_gpu_replays_reference_alan_online_runs(golden_dir, ci):
    """The recorded reference runs, replayed on the GPU with the reference's own uniforms."""
    c = load_case(golden_dir, ci)

    def make(n, scen, p):
        return H.make_gpu(1, n, scen, p, max_obst_neighbors=16)
    env, n, p = setup_env(make, c, lambda e, f, v: e.set(f, v), _lib)
    env.alan_configure(c["actions"])
    steps = c["u"].shape[0]
    for s in range(steps):
        env.alan_step(c["u"][s])
        if s % 10 == 0 or s == steps - 1:
            np.testing.assert_array_equal(env.get(_lib.FLD_POS_X)[0], c["pos"][s][:, 0], err_msg="step %d" % s)
            np.testing.assert_array_equal(env.get(_lib.FLD_POS_Y)[0], c["pos"][s][:, 1])
            np.testing.assert_array_equal(env.get(_lib.FLD_VEL_X)[0], c["vel"][s][:, 0])
            assert_vel_close(env.get(_lib.FLD_VEL_Y)[0], c["vel"][s][:, 1], "step %d" % s)
            np.testing.assert_array_equal(env.get(_lib.FLD_AGENT_DONE)[0], c["done"][s])
        if s % 10 == 0:
            np.testing.assert_allclose(env.get(_lib.FLD_ALAN_WEIGHTS)[0], c["w"][s // 10], rtol=0, atol=1e-13)
            np.testing.assert_array_equal(env.get(_lib.FLD_ALAN_TIMES)[0], c["t"][s // 10])
    np.testing.assert_allclose(env.get(_lib.FLD_ALAN_WEIGHTS)[0], c["w_last"], rtol=0, atol=1e-13)
    times = alan.agents_time(env.get(_lib.FLD_ARRIVE_STEP)[0], env.get(_lib.FLD_AGENT_DONE)[0], p["time_step"], p["max_step"])
    np.testing.assert_allclose(times, c["agents_time"], rtol=0, atol=1e-12)
    assert abs(alan.ttime(times) - float(c["TTime"])) < 1e-12
    env.close()


def test_sim_dropin_run_sim():
    """Collision_Avoidance_Sim drop-in: run_sim() return contract (ALAN_true.py:106-131), one arena and batched."""
    sim = alan.Collision_Avoidance_Sim(numAgents=10, scenario="crowd", visualize=False, seed=2)
    assert sim.max_step == int(600 * 10) and len(sim.online_actions) == 8
    ok, total, tt, mtt = sim.run_sim(1)
    assert ok is True and 0 < total < sim.max_step * sim.timeStep
    assert sim.agents_done == [1] * 10 and max(sim.agents_time) == pytest.approx(total)
    assert tt == sim.TTime and tt > 0 and mtt > 0 and mtt == sim.min_TTime
    # the same world through the oracle's run loop
    p = scenarios.alan_params(10, "crowd")
    e = H.make_oracle(1, 10, "crowd", p, seed=2)
    e.alan_configure(alan.DEFAULT_ACTIONS)
    while not e.get(o.FLD_ARENA_DONE)[0]:
        e.alan_step()
    times = alan.agents_time(e.get(o.FLD_ARRIVE_STEP)[0], e.get(o.FLD_AGENT_DONE)[0], 1 / 60., sim.max_step)
    assert alan.ttime(times) == tt and int(e.get(o.FLD_STEP_COUNT)[0]) * (1 / 60.) == total
    # plain ORCA mode, and a batch of arenas; arena 0 of the batch is the single-arena world
    sim.reset()
    ok0, total0, tt0, _ = sim.run_sim(0)
    assert ok0 and tt0 > 0
    batch = alan.Collision_Avoidance_Sim(numAgents=10, scenario="crowd", seed=2, n_arenas=64)
    okb, totb, ttb, mttb = batch.run_sim(1)
    assert okb.shape == (64,) and okb.all() and ttb[0] == tt and totb[0] == total and mttb[0] == mtt
    assert len(np.unique(totb)) > 8 and (ttb > 0).all()


def test_reset_draws_a_new_world_and_batched_evaluation():
    """reset() re-creates the world like ALAN_true.py:133-139; evaluate_actions() = MCMC_trainer.evaluate_action
    (Train_ALAN_action_space.py:53-66) as one batched run."""
    sim = alan.Collision_Avoidance_Sim(numAgents=8, scenario="crowd", seed=7)
    p0 = sim.vec.get(_lib.FLD_POS_X).copy()
    r1 = sim.run_sim(1)
    sim.reset()
    p1 = sim.vec.get(_lib.FLD_POS_X).copy()
    r2 = sim.run_sim(1)
    sim.reset([(1, 0), (0, 1)])
    assert not np.array_equal(p0, p1) and r1 != r2 and sim.vec.n_actions == 2
    # the three rounds of the trainer's evaluation == arenas 0..2 of one batched handle
    acts = [(1, 0), (0.6, -0.8), (-0.5, 0.86)]
    mean_tt, ok = alan_actions.evaluate_actions(acts, numAgents=8, scenario="crowd", num=3, seed=7)
    seq = alan.Collision_Avoidance_Sim(numAgents=8, scenario="crowd", online_actions=acts, seed=7)
    tts = []
    for r in range(3):
        if r:
            seq.reset(acts)
        tts.append(seq.run_sim(1)[2])
    assert ok == 3 and abs(mean_tt - float(np.mean(tts))) < 1e-12


def test_alan_error_behaviour():
    env = H.make_gpu(2, 4, "crowd", H.scenario_params("crowd", 4))
    with pytest.raises(RuntimeError, match="ca_alan_configure first"):
        env.alan_step()
    with pytest.raises(RuntimeError):
        env.get(_lib.FLD_ALAN_ACTION)                     # no bandit state yet
    with pytest.raises(RuntimeError, match="out of range"):
        env.alan_configure([(1, 0)] * 33)
    with pytest.raises(RuntimeError, match="positive"):
        env.alan_configure([(1, 0), (0, 1)], temp=0.0)
    env.alan_configure([(1, 0), (0, 0)])                  # a zero action vector means "no rotation" (atan2(0, 0) = 0)
    env.alan_step()
    assert env.get(_lib.FLD_ALAN_WEIGHTS).shape == (2, 4, 2)
    with pytest.raises(RuntimeError, match="do not apply"):
        env._call("ca_alan_step", env.h, None, 0, _lib.F_AUTORESET)
    env.close()


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_gpu_replays_reference_orca_episode_loop(golden_dir, ci):
    """run_sim(mode=0) of the reference (recorded in tests/golden/alan_orca.npz) through ca_orca_step."""
    from tests.test_oracle_alan import load_orca_case, replay_orca_episode
    c = load_orca_case(golden_dir, ci)

    def make(n, scen, p):
        return H.make_gpu(1, n, scen, p, max_obst_neighbors=16)
    env, n, p = setup_env(make, c, lambda e, f, v: e.set(f, v), _lib)
    env.set(_lib.FLD_PREF_X, c["pref0"][:, 0]); env.set(_lib.FLD_PREF_Y, c["pref0"][:, 1])
    replay_orca_episode(env, c, _lib, lambda e: e.orca_step(), lambda e, f: e.get(f))
    env.close()


@pytest.mark.parametrize("ci", range(7))
def test_gpu_replays_reference_episodes_that_finish(golden_dir, ci):
    """run_sim(0) and run_sim(1) of the reference recorded to their END (tests/golden/alan_finished.npz: every agent
    arrives and the loop breaks, or the run hits max_step) through ca_orca_step / ca_alan_step: the arena's done flag at
    exactly the reference's last step, total_time, arrival times, TTime, min_TTime, the swapped targets."""
    from tests.test_oracle_alan import load_finished_case, replay_finished_episode
    c = load_finished_case(golden_dir, ci)

    def make(n, scen, p):
        return H.make_gpu(1, n, scen, p, max_obst_neighbors=16)
    env, n, p = setup_env(make, c, lambda e, f, v: e.set(f, v), _lib)
    env.set(_lib.FLD_PREF_X, c["pref0"][:, 0]); env.set(_lib.FLD_PREF_Y, c["pref0"][:, 1])
    if int(c["mode"]) == 1:
        env.alan_configure(c["actions"])
        replay_finished_episode(env, c, _lib, lambda e, s: e.alan_step(c["u"][s]), lambda e, f: e.get(f), p)
        np.testing.assert_allclose(env.get(_lib.FLD_ALAN_WEIGHTS)[0], c["w_last"], rtol=0, atol=1e-13)
        np.testing.assert_array_equal(env.get(_lib.FLD_ALAN_TIMES)[0], c["t_last"])
    else:
        replay_finished_episode(env, c, _lib, lambda e, s: e.orca_step(), lambda e, f: e.get(f), p)
    env.close()


def test_alan_draws_differ_between_episodes_of_an_arena():
    """The built-in draw is keyed by the arena's episode counter as well as its step counter: two
    consecutive episodes of the same arena, started from the same state, explore differently -- and still equal the
    oracle."""
    A, N = 3, 8
    p = H.scenario_params("crowd", N)
    g = H.make_gpu(A, N, "crowd", p, seed=9)
    e = H.make_oracle(A, N, "crowd", p, seed=9)
    g.alan_configure(alan.DEFAULT_ACTIONS); e.alan_configure(alan.DEFAULT_ACTIONS)
    px, py = g.get(_lib.FLD_POS_X).copy(), g.get(_lib.FLD_POS_Y).copy()
    runs = []
    for ep in range(2):
        g.reset(px, py, with_obs=False); e.reset(px, py, flags=0)        # the same start, the next episode
        acts = []
        for s in range(12):
            g.alan_step(); e.alan_step()
            acts.append(g.get(_lib.FLD_ALAN_ACTION).copy())
            H._eq(acts[-1], e.get(o.FLD_ALAN_ACTION), "episode %d step %d actions" % (ep, s))
        runs.append(np.stack(acts))
    H.assert_state_equal(g, e, "two alan episodes")
    assert (runs[0][0] != runs[1][0]).any() and (runs[0] != runs[1]).mean() > 0.3
    g.close()


def test_fused_alan_launch_equals_three_launch_form_and_oracle():
    """Small batches run the bandit INSIDE the four-lanes kernel's launch (one launch per step, one per 256 steps of a
    rollout; csrc/ca_quad.h); CA_ALAN_FUSED=0 keeps the select -> solve -> update launches.  Both against the oracle, bit
    for bit (fp64 weights and times included), single steps with caller-supplied uniforms, with the handle's own draws, and
    as a frozen rollout that ends every arena at its own last step (run_sim, ALAN:106-123)."""
    import os
    A, N = 40, 12
    p = H.scenario_params("circle", N, max_step=90)
    fused = H.make_gpu(A, N, "circle", p, seed=8)
    os.environ["CA_ALAN_FUSED"] = "0"
    try:
        plain = H.make_gpu(A, N, "circle", p, seed=8)
        plain.alan_configure(ACTS9)
    finally:
        del os.environ["CA_ALAN_FUSED"]
    e = H.make_oracle(A, N, "circle", p, seed=8)
    fused.alan_configure(ACTS9); e.alan_configure(ACTS9)
    assert fused.launch_info()["lanes_per_agent"] == (1 if os.environ.get("CA_QUAD") == "0" else 4)   # (the suite is also run with the choice forced)
    rng = np.random.RandomState(6)
    for s in range(12):                                # the uniforms numpy's choice would consume (ALAN:585)
        u = rng.uniform(0, 1, (A, N))
        for env in (fused, plain):
            env.alan_step(u=u, stats=True)
        e.alan_step(u=u, flags=o.F_STATS)
    _assert_alan_equal(fused, e, "fused, given uniforms")
    _assert_alan_equal(plain, e, "three launches, given uniforms")
    for s in range(10):
        for env in (fused, plain):
            env.alan_step(stats=True, with_obs=(s == 9))
        e.alan_step(flags=o.F_STATS | (o.F_OBS if s == 9 else 0))
    _assert_alan_equal(fused, e, "fused, own draws")
    H._eq(fused.get(_lib.FLD_OBS), e.get(o.FLD_OBS), "fused obs")
    sc = (np.arange(A) % 31).astype(np.int32) + fused.get(_lib.FLD_STEP_COUNT)     # the arenas end at different steps
    for env in (fused, plain):
        env.set(_lib.FLD_STEP_COUNT, sc)
    e.set(o.FLD_STEP_COUNT, sc)
    fused.alan_rollout(300, stats=True, freeze=True)     # 256 + 44 steps: two launches
    plain.alan_rollout(300, stats=True, freeze=True)
    for s in range(300):
        e.alan_step(flags=o.F_STATS | o.F_FREEZE)
    _assert_alan_equal(fused, e, "fused rollout")
    _assert_alan_equal(plain, e, "three-launch rollout")
    H.assert_stats_equal(fused, e, "fused rollout")
    assert fused.get(_lib.FLD_ARENA_DONE).all()
    fused.close(); plain.close()


@pytest.mark.parametrize("scenario,A,N,acts,over,one_launch", [
    ("crowd", 30, 16, alan.DEFAULT_ACTIONS, {}, True),                 # one wave, K = 10, 8 actions
    ("circle", 5, 100, ACTS9, {}, True),                               # two waves per arena, 9 actions
    ("crowd", 9, 40, alan.DEFAULT_ACTIONS[:3], dict(max_neighbors=5, neighbor_dist=2.0), True),   # the K = 5 kernel
    ("crowd", 6, 24, [(1, 0), (0, 1)] * 16, {}, False),                # 32 actions: more than the pool holds -> three launches
    # the worlds of the reference's own ALAN runs (ALAN:738-772) that have obstacles:
    ("congested", 7, 50, ACTS9, {}, True),                             # register lines, obstacle lists of 16 (a world of <= 16 edges)
    ("congested", 3, 100, ACTS9, {}, True),                            # ... two waves per arena
    ("deadlock", 7, 50, ACTS9[:2], {}, True),                          # the LDS line table (42 edges), softmax terms in the table
    ("deadlock", 3, 100, ACTS9, {}, True),                             # ... two waves per arena (57 KB of LDS)
    ("blocks", 9, 20, alan.DEFAULT_ACTIONS, {}, True),                 # a world per arena
    ("deadlock", 4, 40, [(1, 0), (0, 1)] * 16, {}, True),              # 32 actions fit the table (2 (K + S) = 52 per lane)
])
def test_alan_inside_the_lane_kernel(scenario, A, N, acts, over, one_launch):
    """Batches on the one-lane-per-agent kernels (forced here with CA_QUAD=0; by default every batch of 1024 or
    more waves) run the bandit inside the solve launch too (csrc/ca_step.h, ALAN instantiation): against the oracle, bit for
    bit, with caller-supplied uniforms, the handle's own draws, statistics, observation and per-arena freezing.  The
    profile shows the form taken: no launch of the small select / update kernels when the bandit is inside the solve."""
    import os
    p = H.scenario_params(scenario, N, max_step=70, **over)
    os.environ["CA_QUAD"] = "0"
    try:
        g = H.make_gpu(A, N, scenario, p, seed=5)
    finally:
        del os.environ["CA_QUAD"]
    e = H.make_oracle(A, N, scenario, p, seed=5)
    g.alan_configure(acts); e.alan_configure(acts)
    assert g.launch_info()["lanes_per_agent"] == 1
    rng = np.random.RandomState(9)
    g.profile(1); g.profile_read()
    for s in range(8):
        u = rng.uniform(0, 1, (A, N))
        g.alan_step(u=u, stats=True); e.alan_step(u=u, flags=o.F_STATS)
    prof = g.profile_read(); g.profile(0)
    assert prof["step_kernel"][0] == 8 and prof["reset_kernels"][0] == (0 if one_launch else 16), prof
    _assert_alan_equal(g, e, "given uniforms")
    sc = (np.arange(A) % 17).astype(np.int32) + g.get(_lib.FLD_STEP_COUNT)     # the arenas end at different steps
    g.set(_lib.FLD_STEP_COUNT, sc); e.set(o.FLD_STEP_COUNT, sc)
    for s in range(50):
        obs = s % 25 == 24
        live = g.get(_lib.FLD_ARENA_DONE) == 0 if obs else None     # (an arena that is frozen already is not observed again by the oracle)
        g.alan_step(stats=True, freeze=True, with_obs=obs); e.alan_step(flags=o.F_STATS | o.F_FREEZE | (o.F_OBS if obs else 0))
        if obs:
            _assert_alan_equal(g, e, "own draws, step %d" % s)
            H._eq(g.get(_lib.FLD_OBS)[live], e.get(o.FLD_OBS)[live], "obs step %d" % s)
            assert live.any()
    g.alan_rollout(40, stats=True, freeze=True)
    for s in range(40):
        e.alan_step(flags=o.F_STATS | o.F_FREEZE)
    _assert_alan_equal(g, e, "rollout")
    H.assert_stats_equal(g, e, "lane ALAN")
    g.close()
