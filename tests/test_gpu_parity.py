"""GPU parity tests: the HIP environment (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: bit-exact (fp32 state, observation, reward, neighbour lists, counters)."""
import os

import numpy as np
import pytest

from collision_avoidance_amd import scenarios
from oracle import oracle as o
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dbg():
    env = H.make_gpu(1, 4, "crowd", H.scenario_params("crowd", 4), seed=12345)
    yield env
    env.close()


def _debug(env, op, inp, out):
    env._call("ca_debug_math", env.h, op, inp.ctypes.data, out.ctypes.data, len(out) if op < 2 or op >= 6 else len(out) // 2)


def test_numerics_contract_sqrt_div(dbg):
    rng = np.random.RandomState(0)
    x = np.abs(rng.standard_normal(1 << 16)).astype(np.float32) * np.float32(10) ** rng.randint(-20, 20, 1 << 16).astype(np.float32)
    out = np.empty_like(x)
    _debug(dbg, 0, x, out)
    np.testing.assert_array_equal(out, np.sqrt(x))
    ab = (rng.standard_normal((1 << 16, 2)) * 10.0 ** rng.randint(-15, 15, (1 << 16, 2))).astype(np.float32)
    out = np.empty(1 << 16, np.float32)
    _debug(dbg, 1, ab, out)
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(out, ab[:, 0] / ab[:, 1])


def test_numerics_contract_division_in_range(dbg):
    """csrc/ca_math.h div_ir: the compiler's correctly rounded fp32 division minus its range scaling and fix-up instructions,
    used where the operands are in range by construction (LP1 clips, the reciprocals of the half-planes).  It must return the
    bits of IEEE division -- the oracle's host division -- for every pair of its domain: 2e8 pairs here (random mantissas with
    exponents over the whole domain 2^-60 .. 2^60, numerators that are exact multiples, quotients next to a rounding
    boundary, a zero numerator); tools/diag/div_exhaustive.py runs 2e9."""
    rng = np.random.RandomState(7)
    n = 1 << 22
    for rnd in range(48):
        mant = lambda: (rng.randint(0, 1 << 23, n).astype(np.uint32) | np.uint32(0x3F800000)).view(np.float32)    # [1, 2)
        ea, eb = rng.randint(-60, 61, n), rng.randint(-60, 61, n)
        if rnd % 4 == 1:      # the ranges of the LP1 clip: den in (1e-5, 1], num in [1e-21, 240]
            ea, eb = rng.randint(-70, 8, n), rng.randint(-17, 1, n)
        b = np.ldexp(mant(), eb).astype(np.float32) * rng.choice(np.float32([-1, 1]), n)
        a = np.ldexp(mant(), ea).astype(np.float32) * rng.choice(np.float32([-1, 1]), n)
        if rnd % 4 == 2:      # quotients that sit next to a rounding boundary: a = fl(q * b) for a random q, then one ulp either way
            q = np.ldexp(mant(), rng.randint(-20, 21, n)).astype(np.float32)
            a = np.nextafter((q * b).astype(np.float32), np.float32(np.inf) * rng.choice(np.float32([-1, 1]), n)).astype(np.float32)
        if rnd % 4 == 3:      # exact quotients and a zero numerator
            a = (b * rng.randint(-4096, 4097, n).astype(np.float32)).astype(np.float32)
        ab = np.ascontiguousarray(np.stack([a, b], 1))
        out = np.empty(n, np.float32)
        _debug(dbg, 6, ab, out)
        with np.errstate(all="ignore"):
            ref = (a / b).astype(np.float32)
        bad = out.view(np.uint32) != ref.view(np.uint32)
        assert not bad.any(), (rnd, int(bad.sum()), a[bad][:3], b[bad][:3], out[bad][:3], ref[bad][:3])


def test_numerics_contract_sqrt_in_range(dbg):
    """csrc/ca_math.h sqrt_ir: the compiler's correctly rounded square root without its scaling of tiny arguments and its class
    fix-up; domain x = 0 or x >= 2^-96.  The bits of IEEE sqrt for 1.3e8 arguments: every exponent of the domain with random
    mantissas, the squares of random numbers and their two neighbours (the rounding boundaries), 0 and +inf."""
    rng = np.random.RandomState(8)
    n = 1 << 22
    for rnd in range(32):
        m = (rng.randint(0, 1 << 23, n).astype(np.uint32) | np.uint32(0x3F800000)).view(np.float32)
        x = np.ldexp(m, rng.randint(-96, 120, n)).astype(np.float32)
        if rnd % 4 == 1:
            y = np.ldexp(m, rng.randint(-40, 40, n)).astype(np.float32)
            x = (y * y).astype(np.float32)
            x = np.where(rng.randint(0, 3, n) == 0, x, np.nextafter(x, np.float32(np.inf) * rng.choice(np.float32([-1, 1]), n))).astype(np.float32)
        if rnd % 4 == 2:      # the range of the discriminants and squared lengths of the solver
            x = np.ldexp(m, rng.randint(-54, 16, n)).astype(np.float32)
        x[:4] = [0.0, np.inf, 1.0, 2.0 ** -96]
        out = np.empty(n, np.float32)
        _debug(dbg, 7, np.ascontiguousarray(x), out)
        ref = np.sqrt(x).astype(np.float32)
        bad = out.view(np.uint32) != ref.view(np.uint32)
        assert not bad.any(), (rnd, int(bad.sum()), x[bad][:3], out[bad][:3], ref[bad][:3])


def test_numerics_contract_fp64_helpers(dbg):
    rng = np.random.RandomState(1)
    a = np.concatenate([rng.uniform(-np.pi, np.pi, 20000), rng.uniform(-1e4, 1e4, 20000),
                        np.float32(rng.uniform(-np.pi, np.pi, 20000)).astype(np.float64), [0.0, np.pi / 4, -np.pi / 4]])
    out = np.empty(2 * len(a))
    _debug(dbg, 2, a, out)
    ref = np.array([o.sincos64(v) for v in a]).reshape(-1)
    np.testing.assert_array_equal(out, ref)
    f = rng.uniform(-50, 50, (30000, 4)).astype(np.float32)
    f[:10, 2:] = f[:10, :2]          # zero vector -> (1, 0)
    out = np.empty(2 * len(f))
    _debug(dbg, 3, f, out)
    ref = np.array([o.pref_dir64(*row) for row in f]).reshape(-1)
    np.testing.assert_array_equal(out, ref)


def test_numerics_contract_rng(dbg):
    rng = np.random.RandomState(2)
    u = rng.randint(0, 2 ** 31, (5000, 4)).astype(np.uint32)
    out = np.empty(2 * len(u))
    _debug(dbg, 4, u, out)
    seed = 12345
    ref = []
    for g, i, p, s in u:
        w = o.philox4x32((int(g), int(i), int(p), int(s)), (seed & 0xffffffff, seed >> 32))
        ref += [(((w[0] >> 5) << 26) | (w[1] >> 6)) * 2.0 ** -53, (((w[2] >> 5) << 26) | (w[3] >> 6)) * 2.0 ** -53]
    np.testing.assert_array_equal(out, np.array(ref))


CASES = [
    # name,     A,  N,  scenario, overrides,                                 steps, check_every
    ("c2like", 48, 16, "crowd", dict(neighbor_dist=1.5, max_neighbors=5), 240, 40),
    ("c3like", 24, 64, "crowd", dict(), 200, 50),
    ("c5like", 2, 512, "crowd", dict(), 60, 20),
    ("circle8", 5, 8, "circle", dict(), 400, 50),
    ("circle100", 2, 100, "circle", dict(), 150, 50),
    ("doorway", 32, 10, "doorway", dict(), 500, 50),
    ("odd_n", 7, 23, "crowd", dict(max_neighbors=7, neighbor_dist=3.0), 150, 50),
    ("k16", 3, 40, "crowd", dict(max_neighbors=16, neighbor_dist=6.0), 100, 50),
    ("k0", 3, 12, "crowd", dict(max_neighbors=0), 50, 25),
    ("single", 4, 1, "crowd", dict(), 50, 25),
    # the other ALAN worlds: convex blocks, slanted funnels, a two-way tube, an incoming platoon
    ("deadlock", 6, 20, "deadlock", dict(), 700, 100),
    ("incoming", 6, 17, "incoming", dict(), 400, 100),
    ("congested", 6, 24, "congested", dict(), 500, 100),
    ("blocks", 6, 12, "blocks", dict(), 500, 100),
    # the largest workgroup shapes: 1024 lanes (register lines) and the LDS line table with K = 16
    ("n1000", 1, 1000, "crowd", dict(), 12, 6),
    ("n250k16", 2, 250, "crowd", dict(max_neighbors=16, neighbor_dist=4.0), 20, 10),
    # the uniform-grid neighbour scan (>= 192 agents per arena): exact distance ties by symmetry (circle), a world
    # whose bounding box is not the arena (incoming platoon), a small neighbour range (many cells)
    ("grid_circle", 2, 256, "circle", dict(), 40, 10),
    ("grid_incoming", 2, 226, "incoming", dict(), 60, 20),
    ("grid_smallrange", 2, 400, "crowd", dict(neighbor_dist=1.5, max_neighbors=5), 40, 10),
    # ... with helper lanes in the scan (192-512 agents): list shorter than the register array (K = 7 of 10, 3 of 5),
    # an arena that fills the 512-lane shape exactly, one just above the threshold
    ("grid_k7", 2, 300, "crowd", dict(neighbor_dist=4.0, max_neighbors=7), 30, 10),
    ("grid_k3", 3, 200, "crowd", dict(neighbor_dist=2.0, max_neighbors=3), 30, 10),
    ("grid_512", 1, 512, "crowd", dict(), 20, 10),
    ("grid_192", 3, 192, "circle", dict(), 30, 10),
    # SURVEY 8d bench variants: rejection-sampled non-overlapping starts
    ("separated", 12, 64, "crowd_separated", dict(), 150, 50),
]


@pytest.mark.parametrize("name,A,N,scenario,over,steps,every", CASES, ids=[c[0] for c in CASES])
def test_orca_rollout_bit_exact(name, A, N, scenario, over, steps, every):
    p = H.scenario_params(scenario, N, **over)
    gpu = H.make_gpu(A, N, scenario, p, seed=3)
    orc = H.make_oracle(A, N, scenario, p, seed=3)
    H.assert_state_equal(gpu, orc, name + " init", lists=False)
    for s in range(steps):
        last = (s + 1) % every == 0 or s < 3
        gpu.orca_step(with_obs=last, stats=True)
        orc.orca_step(flags=(o.F_OBS if last else 0) | o.F_STATS)
        if last:
            H.assert_state_equal(gpu, orc, "%s step %d" % (name, s), obs=True)
    H.assert_stats_equal(gpu, orc, name)
    gpu.close()


@pytest.mark.parametrize("scenario,N", [("deadlock", 20), ("deadlock", 30), ("blocks", 12), ("doorway", 10), ("congested", 24),
                                        ("incoming", 17), ("crowd", 16), ("circle", 12)])
def test_reference_worlds_keep_every_obstacle_edge_in_range(scenario, N):
    """RVO2 keeps EVERY obstacle edge in range (env.py:249, 301-318 read them all); the obstacle-neighbour list here has a
    capacity (16, the default for worlds with that many edges): in the reference's own seven worlds (env.py:77-123,
    ALAN:175-457) no agent-step may lose an edge -- with a capacity of 8 the two-way tube of "deadlock" dropped the
    farthest edges in 0.6 % of the agent-steps (profiles/r02_soak_parity.txt)."""
    p = H.scenario_params(scenario, N)
    gpu = H.make_gpu(8, N, scenario, p, seed=3)
    orc = H.make_oracle(8, N, scenario, p, seed=3)
    gpu.rollout(1500, stats=True)
    orc.rollout(1500, flags=o.F_STATS, n_threads=8)
    H.assert_state_equal(gpu, orc, scenario)
    H.assert_stats_equal(gpu, orc, scenario)
    assert gpu.stats()["obst_overflow"] == 0 and orc.stats()["obst_overflow"] == 0


def test_obs_adversarial_geometry():
    """The observation kernel culls rays by the angular span of each segment; the culling must be a
    superset of what the exact test accepts.  Neighbours are placed exactly on ray directions, on
    the octagon's in/circum-radius, inside the octagon and almost on top of the agent."""
    from collision_avoidance_amd import _lib
    N, dists = 4, [1e-4, 0.03, 0.25, 0.46193975, 0.4619398, 0.5, 0.50000006, 0.7, 1.0, 1.4999, 2.0]
    cases = []
    for k in range(32):                      # direction: every ray and every half-ray
        for d in dists:
            ang = -k * (2 * np.pi / 32)
            cases.append((d * np.cos(ang), d * np.sin(ang)))
    rng = np.random.RandomState(3)
    A = len(cases)
    p = H.scenario_params("crowd", N, neighbor_dist=3.0, max_neighbors=3)
    polys = [[(-50.0, -0.2), (50.0, -0.2)]]  # a 2-vertex wall just below: passes close to the origin
    gpu = H.make_gpu(A, N, "crowd", p, seed=2, polys=polys)
    orc = H.make_oracle(A, N, "crowd", p, seed=2, polys=polys)
    px = np.zeros((A, N), np.float32); py = np.zeros((A, N), np.float32)
    for a, (dx, dy) in enumerate(cases):
        px[a] = [10.0, 10.0 + dx, 10.0 - 2 * dx + 0.9, 10.0 + 1.3]
        py[a] = [0.0, dy, -2 * dy + 0.3, 1.1]
    gxy = rng.uniform(-20, 20, (2, A, N)).astype(np.float32)
    gxy[0, ::3] = px[::3] + 5.0; gxy[1, ::3] = py[::3]      # goal straight along +x: frame unrotated
    vel = rng.uniform(-0.7, 0.7, (2, A, N)).astype(np.float32)
    for env, F in ((gpu, _lib), (orc, o)):
        env.set(F.FLD_POS_X, px); env.set(F.FLD_POS_Y, py)
        env.set(F.FLD_VEL_X, vel[0]); env.set(F.FLD_VEL_Y, vel[1])
        env.set(F.FLD_GOAL_X, gxy[0]); env.set(F.FLD_GOAL_Y, gxy[1])
    for s in range(3):
        gpu.orca_step(with_obs=True, no_done=True)
        orc.orca_step(flags=o.F_OBS | o.F_NODONE)
        H.assert_state_equal(gpu, orc, "adversarial step %d" % s, obs=True)
        # put the agents back so that every step probes the crafted geometry: reset = new positions
        # + observation from the neighbour lists of the last step (env.py:461-488)
        gpu.reset(px, py)
        orc.reset(px, py, flags=o.F_OBS)
        H.assert_state_equal(gpu, orc, "adversarial reset %d" % s, obs=True)
    assert np.abs(gpu.get(_lib.FLD_OBS)).max() > 0
    gpu.close()


def test_configurations_that_do_not_fit_are_refused():
    """K > 10 forces the LDS line table: (K + S) x 16 B per lane must fit 160 KiB with the arena in one
    workgroup.  The library must say so instead of failing inside HIP."""
    p = H.scenario_params("crowd", 700, max_neighbors=16)
    with pytest.raises(RuntimeError, match="LDS"):
        H.make_gpu(1, 700, "crowd", p)
    with pytest.raises(RuntimeError, match="out of range"):
        H.make_gpu(1, 1025, "crowd", H.scenario_params("crowd", 1025))


def test_regoal_and_rollout_call():
    N = 16
    p = scenarios.bench_params(N, 1.5, 5)
    gpu = H.make_gpu(64, N, "crowd", p, seed=9)
    orc = H.make_oracle(64, N, "crowd", p, seed=9)
    for chunk in range(6):
        gpu.rollout(250, stats=True)
        orc.rollout(250, flags=o.F_STATS)
        H.assert_state_equal(gpu, orc, "regoal chunk %d" % chunk)
    H.assert_stats_equal(gpu, orc, "regoal")
    assert gpu.stats()["goals_reached"] > 0
    gpu.close()


@pytest.mark.parametrize("scenario,A,N", [("doorway", 16, 10), ("crowd", 12, 64), ("crowd", 40, 16)])
def test_step_with_actions_bit_exact(scenario, A, N):
    p = H.scenario_params(scenario, N)
    if scenario == "crowd":
        p["reward_scale"] = 0.3
    gpu = H.make_gpu(A, N, scenario, p, seed=5)
    orc = H.make_oracle(A, N, scenario, p, seed=5)
    rng = np.random.RandomState(5)
    gpu.reset(); orc.reset()
    H.assert_state_equal(gpu, orc, "reset", obs=True)
    for s in range(160):
        act = rng.uniform(-np.pi, np.pi, (A, N)).astype(np.float32) * (0.3 if s % 2 else 1.0)
        gpu.step(act, stats=True)
        orc.step(act, flags=o.F_OBS | o.F_STATS)
        if s % 20 == 0 or s < 3:
            H.assert_state_equal(gpu, orc, "%s step %d" % (scenario, s), obs=True, reward=True)
    H.assert_state_equal(gpu, orc, "final", obs=True, reward=True)
    H.assert_stats_equal(gpu, orc, scenario)
    gpu.close()


def test_autoreset_and_explicit_reset():
    A, N = 24, 6
    p = H.scenario_params("doorway", N, max_step=60)
    gpu = H.make_gpu(A, N, "doorway", p, seed=8)
    orc = H.make_oracle(A, N, "doorway", p, seed=8)
    gpu.reset(); orc.reset()
    rng = np.random.RandomState(8)
    for s in range(200):
        act = rng.uniform(-0.5, 0.5, (A, N)).astype(np.float32)
        gpu.step(act, stats=True, autoreset=True)
        orc.step(act, flags=o.F_OBS | o.F_STATS | o.F_AUTORESET)
        if s % 10 == 9:
            H.assert_state_equal(gpu, orc, "autoreset step %d" % s, obs=True, reward=True)
    assert gpu.stats()["episodes"] >= 3 * A
    H.assert_stats_equal(gpu, orc, "autoreset")
    px = rng.uniform(5, 10, (A, N)).astype(np.float32); py = rng.uniform(0, 10, (A, N)).astype(np.float32)
    gpu.reset(px, py); orc.reset(px, py)
    H.assert_state_equal(gpu, orc, "explicit reset", obs=True)
    gpu.close()


def test_sharding_invariance():
    """Arena g computes the same thing whichever shard owns it (every draw keyed by the GLOBAL arena id; run_rllib.py:108 replicates
    environments over workers the same way): a handle of 8 arenas against two handles of 4 -- ORCA rollout, then full steps with
    the same actions, statistics and auto-reset on -- compared field by field as bit patterns: the whole state, the neighbour
    lists, reward, observation and the per-arena counters."""
    from collision_avoidance_amd import _lib
    N = 16
    p = scenarios.bench_params(N, 1.5, 5)
    whole = H.make_gpu(8, N, "crowd", p, seed=4)
    parts = [H.make_gpu(4, N, "crowd", p, seed=4, arena_offset=off) for off in (0, 4)]
    rng = np.random.RandomState(11)
    acts = rng.uniform(-1, 1, (40, 8, N)).astype(np.float32)
    for e, sl in [(whole, slice(0, 8)), (parts[0], slice(0, 4)), (parts[1], slice(4, 8))]:
        e.rollout(700, stats=True)
        for t in range(40):
            e.step(acts[t, sl], stats=True, autoreset=True)
    names = ["POS_X", "POS_Y", "VEL_X", "VEL_Y", "PREF_X", "PREF_Y", "GOAL_X", "GOAL_Y", "GOAL2_X", "GOAL2_Y", "REWARD", "AGENT_DONE",
             "ARRIVE_STEP", "NB_COUNT", "OBST_COUNT", "OBST_IDX", "OBS", "STEP_COUNT", "ARENA_DONE", "EPISODE", "REGOAL_COUNT", "ARENA_STATS"]
    for nm in names:
        f = getattr(_lib, "FLD_" + nm)
        H._eq(whole.get(f), np.concatenate([e.get(f) for e in parts]), "sharding " + nm)
    wc, wi = whole.neighbor_lists()                      # (entries beyond the count are not defined)
    pi = np.concatenate([e.neighbor_lists()[1] for e in parts])
    m = np.arange(wi.shape[2])[None, None, :] < wc[:, :, None]
    H._eq(np.where(m, wi, -1), np.where(m, pi, -1), "sharding NB_IDX")
    tot = whole.stats()
    for k in ("agent_steps", "episodes", "collisions", "obst_collisions", "goals_reached", "obst_overflow"):
        assert tot[k] == sum(e.stats()[k] for e in parts), k
    assert tot["goals_reached"] > 0 and tot["agent_steps"] == 8 * N * 740
    for e in [whole] + parts:
        e.close()


def _golden_replay_gpu(golden_dir, name):
    from collision_avoidance_amd import _lib
    from tests.test_oracle_golden import flip_margins, FLIP_MARGIN
    margins = flip_margins(golden_dir, name)   # fp64 oracle replay: how close each recorded ray is to flipping

    def count_bad(err, key):
        bad = err.reshape(n, 16, 4).max(axis=2) > 3e-5
        if bad.any():   # fp32 vs the reference's fp64: only rays that graze a segment end / tie two hits may differ
            assert (margins[key][bad] < FLIP_MARGIN).all(), (key, np.argwhere(bad).tolist(), margins[key][bad])
        return int(bad.sum())
    g = np.load(os.path.join(golden_dir, name))
    n = int(g["n_agents"])
    env = H.make_gpu(1, n, "doorway", scenarios.env_params(), max_obst_neighbors=16)
    assert (env.get(_lib.FLD_GOAL2_X) == -10.0).all() and (env.get(_lib.FLD_GOAL2_Y) == 5.0).all()   # env.py:361: the doorway
    env.set(_lib.FLD_POS_X, g["pos0"][:, 0]); env.set(_lib.FLD_POS_Y, g["pos0"][:, 1])               # world's own retarget
    env.set(_lib.FLD_VEL_X, g["vel0"][:, 0]); env.set(_lib.FLD_VEL_Y, g["vel0"][:, 1])
    env.set(_lib.FLD_PREF_X, g["pref0"][:, 0]); env.set(_lib.FLD_PREF_Y, g["pref0"][:, 1])
    env.set(_lib.FLD_GOAL_X, g["tgt0"][:, 0]); env.set(_lib.FLD_GOAL_Y, g["tgt0"][:, 1])
    obs_at = {int(s): k for k, s in enumerate(g["obs_steps"])}
    reset_at = {int(s): k for k, s in enumerate(g["reset_steps"])}
    bad = tot = 0
    for s in range(len(g["kind"])):
        if s in reset_at:
            k = reset_at[s]
            ob = env.reset(g["reset_pos"][k][:, 0], g["reset_pos"][k][:, 1])
            err = np.abs(ob[0].astype(np.float64) - g["reset_obs"][k])
            bad += count_bad(err, ("reset", k)); tot += n * 16
        if g["kind"][s] == 1:
            ob = env.orca_step(with_obs=True, no_done=True)
        else:
            ob, rew, done, _ = env.step(g["actions"][s])
            np.testing.assert_allclose(rew[0], g["reward"][s], rtol=0, atol=1e-6)
            assert bool(done[0]) == bool(g["done_all"][s])
        st = env.state()
        # trajectories: the north-star tolerance is 1e-4; the reference run here shares the ORCA
        # arithmetic, so they are in fact identical
        np.testing.assert_array_equal(st["pos_x"][0], g["pos"][s][:, 0], err_msg="step %d" % s)
        np.testing.assert_array_equal(st["pos_y"][0], g["pos"][s][:, 1])
        np.testing.assert_array_equal(st["vel_x"][0], g["vel"][s][:, 0])
        np.testing.assert_array_equal(st["vel_y"][0], g["vel"][s][:, 1])
        np.testing.assert_array_equal(st["pref_x"][0], g["pref"][s][:, 0])
        np.testing.assert_array_equal(st["pref_y"][0], g["pref"][s][:, 1])
        np.testing.assert_array_equal(st["agent_done"][0], g["agents_done"][s])
        np.testing.assert_array_equal(st["goal_x"][0], g["tgt"][s][:, 0])       # the (-10, 5) retarget, kept by reset()
        np.testing.assert_array_equal(st["goal_y"][0], g["tgt"][s][:, 1])
        if "step_count" in g.files:                                              # the counter, its cap, its reset
            assert int(st["step_count"][0]) == int(g["step_count"][s]), s
        if s in obs_at:
            err = np.abs(ob[0].astype(np.float64) - g["obs"][obs_at[s]])
            bad += count_bad(err, ("step", s)); tot += n * 16
    env.close()
    return bad, tot


@pytest.mark.parametrize("name", ["env_doorway_n10.npz", "env_doorway_n6_dense.npz", "env_doorway_n6_episode.npz",
                                  "env_doorway_n4_all_done.npz"])
def test_gpu_replays_reference_env_loop_golden(golden_dir, name):
    """Golden runs of the reference's own env loop (reset / step / orca_step / _get_obs / done_test around doStep) --
    recorded with the oracle's ORCA standing in for the absent rvo2 module (tests/golden/make_golden.py), so this pins the
    Python loops and the laser observation, not the ORCA solver itself."""
    bad, tot = _golden_replay_gpu(golden_dir, name)
    assert bad <= 2, (bad, tot)        # (observed: 0 of ~46 000 rays; each one that differs must be within FLIP_MARGIN of flipping: count_bad)


@pytest.mark.parametrize("name", ["env_doorway_n6_episode.npz", "env_doorway_n4_all_done.npz"])
def test_gpu_autoreset_at_the_recorded_end_of_an_episode(golden_dir, name):
    """The recorded episode again, but with the reset taken INSIDE the step that ends it (CA_F_AUTORESET) instead of by
    the caller: up to that step the run is the reference's; the step itself reports '__all__' and leaves what the
    reference's reset() leaves (env.py:461-488) -- counter and done flags zeroed, the swapped (-10, 5) targets KEPT, new
    positions inside the spawn box (drawn by the handle's own stream: not the fixture's) -- and equals the oracle
    doing the same."""
    from collision_avoidance_amd import _lib
    g = np.load(os.path.join(golden_dir, name))
    n = int(g["n_agents"])
    r = int(g["reset_steps"][0])
    env = H.make_gpu(1, n, "doorway", scenarios.env_params(), max_obst_neighbors=16, seed=3)
    orc = H.make_oracle(1, n, "doorway", scenarios.env_params(), max_obst_neighbors=16, seed=3)
    for e, F in ((env, _lib), (orc, o)):
        for f, v in (("POS_X", g["pos0"][:, 0]), ("POS_Y", g["pos0"][:, 1]), ("VEL_X", g["vel0"][:, 0]), ("VEL_Y", g["vel0"][:, 1]),
                     ("PREF_X", g["pref0"][:, 0]), ("PREF_Y", g["pref0"][:, 1]), ("GOAL_X", g["tgt0"][:, 0]), ("GOAL_Y", g["tgt0"][:, 1])):
            e.set(getattr(F, "FLD_" + f), v)
    for s in range(r):
        _, _, done, _ = env.step(g["actions"][s], autoreset=True, stats=True)
        orc.step(g["actions"][s], flags=o.F_OBS | o.F_STATS | o.F_AUTORESET)
        assert bool(done[0]) == bool(g["done_all"][s]), s
        if s < r - 1:
            st = env.state()
            np.testing.assert_array_equal(st["pos_x"][0], g["pos"][s][:, 0], err_msg="step %d" % s)
            np.testing.assert_array_equal(st["agent_done"][0], g["agents_done"][s])
    st = env.state()
    arrived = g["agents_done"][r - 1] == 1
    assert int(st["step_count"][0]) == 0 and st["agent_done"][0].max() == 0
    assert (st["goal_x"][0][arrived] == -10.0).all() and (st["goal_y"][0] == 5.0).all() and (st["goal_x"][0][~arrived] == 1.0).all()
    assert (st["pos_x"][0] >= 5.0).all() and (st["pos_x"][0] <= 10.0).all() and (st["pos_y"][0] >= 0).all() and (st["pos_y"][0] <= 10.0).all()
    assert int(env.get(_lib.FLD_EPISODE)[0]) == 1 and env.stats()["episodes"] == 1
    H.assert_state_equal(env, orc, name + " after the autoreset", obs=True, reward=True)
    for s in range(r, r + 20):                         # and the next episode starts like any other
        env.step(g["actions"][s], autoreset=True, stats=True)
        orc.step(g["actions"][s], flags=o.F_OBS | o.F_STATS | o.F_AUTORESET)
    H.assert_state_equal(env, orc, name + " 20 steps into the next episode", obs=True, reward=True)
    H.assert_stats_equal(env, orc, name)
    env.close()


def test_dropin_env_api():
    from collision_avoidance_amd.envs import Collision_Avoidance_Env, CollisionAvoidanceEnv
    assert CollisionAvoidanceEnv is Collision_Avoidance_Env
    env = Collision_Avoidance_Env(numAgents=6)
    ob = env.reset()
    assert sorted(ob) == ['agent_%d' % i for i in range(6)] and all(len(v) == 64 for v in ob.values())
    assert all(v == [0.0] * 64 for v in ob.values())        # no doStep yet: empty neighbour lists
    rng = np.random.RandomState(0)
    for _ in range(30):
        act = {'agent_%d' % i: rng.uniform(-1, 1, 1).astype(np.float32) for i in range(6)}
        ob2, rew, dones, infos = env.step(act)
        assert ob2 is env.gym_obs and rew is env.gym_rewards and dones is env.gym_dones
    assert set(dones) == {'__all__'} | set(ob) and dones['__all__'] is False
    assert all(isinstance(v, float) and -1.0 - 1e-5 <= v <= 1.0 + 1e-5 for v in rew.values())
    assert env.step_count == 30
    with pytest.raises(KeyError):
        env.step({'agent_0': [0.0]})
    assert env.orca_step((0, 0)) is None and env.step_count == 30
    assert env.seed(3) == [3]
    assert env.action_space.shape == (1,) and env.observation_space.shape == (64,)
    env.close()


FULL = [  # BASELINE.json configs[1], [2], [4] at their full sizes
    ("C2", 1024, 16, 1.5, 5, 120),
    ("C3", 4096, 64, 5.0, 10, 100),
    ("C5", 256, 512, 5.0, 10, 40),
]


@pytest.mark.parametrize("name,A,N,nd,K,steps", FULL, ids=[c[0] for c in FULL])
def test_full_size_properties(name, A, N, nd, K, steps):
    """BASELINE configs C2 (1024 x 16), C3 (4096 x 64), C5 (256 x 512) at full size: EVERY arena against the oracle bit
    for bit (the oracle steps the whole batch on a thread pool: ORCA rollout with statistics, then full steps with
    actions, rewards and the observation -- state, lists, reward, observation, per-arena counters), so that what is
    keyed by the workgroup index (the observation's XCD remap, arenas per workgroup, the block order) is covered at the
    bench shapes; plus size-independent properties over the whole batch."""
    from collision_avoidance_amd import _lib
    nt = min(16, os.cpu_count() or 1)
    p = scenarios.bench_params(N, nd, K)
    env = H.make_gpu(A, N, "crowd", p, seed=0, use_torch=False)
    orc = H.make_oracle(A, N, "crowd", p, seed=0)
    env.rollout(steps, stats=True)
    orc.rollout(steps, flags=o.F_STATS, n_threads=nt)
    H.assert_state_equal(env, orc, name + " after the ORCA rollout (all arenas)")
    vx, vy = env.get(_lib.FLD_VEL_X), env.get(_lib.FLD_VEL_Y)
    assert np.isfinite(vx).all() and np.isfinite(vy).all()
    # |v| <= maxSpeed up to fp32 cancellation in LP1 (t = -dp +- sqrt(disc) with |point| ~ 1/dt when
    # agents overlap): the oracle shows the same overshoot, bit for bit
    assert np.hypot(vx, vy).max() <= 1.0 + 2e-2
    cnt, idx = env.neighbor_lists()
    assert cnt.max() <= K and cnt.min() >= 0
    px, py = env.get(_lib.FLD_POS_X), env.get(_lib.FLD_POS_Y)
    a = np.arange(A)[:, None, None]
    valid = np.arange(K)[None, None, :] < cnt[:, :, None]
    idx = np.where(valid, idx, 0)                                              # entries beyond the count are unspecified
    d = np.hypot(px[a, idx] - px[:, :, None], py[a, idx] - py[:, :, None])
    assert not np.any(valid & (idx == np.arange(N)[None, :, None]))            # never itself
    assert np.all((idx >= 0) | ~valid) and np.all((idx < N) | ~valid)
    # lists are those of the last doStep (pre-update positions); both agents moved <= maxSpeed*dt since
    assert np.all(d[valid] < nd + 2 * 1.02 / 60 + 1e-4)
    # sorted by the distance of the step that built them: after one step of motion still ascending within 4 v dt
    dd = np.where(valid, d, np.inf)
    both = valid[:, :, 1:] & valid[:, :, :-1]
    assert np.all((dd[:, :, 1:] - dd[:, :, :-1])[both] >= -4 * 1.02 / 60 - 1e-4)
    rng = np.random.RandomState(5)
    for s in range(3):                                                     # full steps: actions in, obs out
        act = rng.uniform(-0.5, 0.5, (A, N)).astype(np.float32)
        ob, rew, done, _ = env.step(act, stats=True)
        orc.step_mt(act, flags=o.F_OBS | o.F_STATS, n_threads=nt)
    assert ob.shape == (A, N, 64) and np.isfinite(ob).all() and np.abs(ob[..., :2]).max() <= nd + 1e-4
    assert np.all(rew <= 1.0 + 2e-2) and not done.any()
    H.assert_state_equal(env, orc, name + " after three full steps (all arenas)", obs=True, reward=True)
    H.assert_stats_equal(env, orc, name)
    st = env.stats()
    assert st["agent_steps"] == A * N * (steps + 3) and st["obst_overflow"] == 0
    gs, es = env.get(_lib.FLD_ARENA_STATS), orc.get(o.FLD_ARENA_STATS)
    np.testing.assert_array_equal(gs[:, [0, 1, 2, 3, 4, 6, 7]], es[:, [0, 1, 2, 3, 4, 6, 7]])
    np.testing.assert_allclose(gs[:, 5].copy().view(np.float64), es[:, 5].copy().view(np.float64), rtol=1e-12, atol=1e-12)
    # rays that hit nothing are exactly zero; a hit lies inside the sensor range and carries the owner's velocity
    hit = (ob[..., 0::4] != 0) | (ob[..., 1::4] != 0)
    assert np.all(ob[..., 2::4][~hit] == 0) and np.all(ob[..., 3::4][~hit] == 0)
    assert np.hypot(ob[..., 0::4], ob[..., 1::4]).max() <= nd * (1 + 1e-6)
    env.close()


SETTLED = [("C3-shaped", 64, 64, 5.0, 10, 10000), ("C5-shaped", 8, 512, 5.0, 10, 10000), ("C2-shaped", 256, 16, 1.5, 5, 10000)]


@pytest.mark.parametrize("name,A,N,nd,K,steps", SETTLED, ids=[c[0] for c in SETTLED])
def test_parity_in_the_settled_regime_the_bench_times(name, A, N, nd, K, steps):
    """bench.py times the crowd AFTER >= 8 000 warm-up steps: contracted, full neighbour lists, ~14 overlapping pairs
    per arena-step against ~2 in the first few hundred (round 4's one parity bug lived in a regime no short run
    reached).  The bench's own loop -- the same 16-entry action pool, one full step per entry, statistics on, no
    reset -- for 10 000 steps on a C3-, C5- and C2-shaped batch, every arena against the oracle bit for bit: state,
    lists, counters; then full steps with the observation."""
    from collision_avoidance_amd import _lib
    nt = min(16, os.cpu_count() or 1)
    p = scenarios.bench_params(N, nd, K)
    env = H.make_gpu(A, N, "crowd", p, seed=0, use_torch=False)
    orc = H.make_oracle(A, N, "crowd", p, seed=0)
    pool = np.random.RandomState(1234).uniform(-0.5, 0.5, (16, A, N)).astype(np.float32)
    early = None
    for s in range(steps):
        env._call("ca_step_host", env.h, pool[s % 16].ctypes.data, _lib.F_STATS)         # no observation, nothing copied back
        if s == 599:
            early = env.stats()["collisions"] / 600.0 / A
    orc.rollout_mt(steps, pool, flags=o.F_STATS, n_threads=nt)
    H.assert_state_equal(env, orc, name + " after %d steps of the bench loop (all arenas)" % steps, reward=True)
    H.assert_stats_equal(env, orc, name)
    late0 = env.stats()["collisions"]
    for s in range(steps, steps + 200):                                                   # the regime itself, measured
        env._call("ca_step_host", env.h, pool[s % 16].ctypes.data, _lib.F_STATS)
    orc.rollout_mt(200, np.roll(pool, -(steps % 16), axis=0), flags=o.F_STATS, n_threads=nt)
    late = (env.stats()["collisions"] - late0) / 200.0 / A
    assert N < 64 or late > 2 * early, (early, late)        # the dense shapes have settled into the contact regime
    for s in range(steps + 200, steps + 204):                                             # full steps: observation out
        ob, rew, done, _ = env.step(pool[s % 16], stats=True)
        orc.step_mt(pool[s % 16], flags=o.F_OBS | o.F_STATS, n_threads=nt)
    H.assert_state_equal(env, orc, name + " full steps in the settled regime", obs=True, reward=True)
    H.assert_stats_equal(env, orc, name)
    gs, es = env.get(_lib.FLD_ARENA_STATS), orc.get(o.FLD_ARENA_STATS)
    np.testing.assert_array_equal(gs[:, [0, 1, 2, 3, 4, 6, 7]], es[:, [0, 1, 2, 3, 4, 6, 7]])
    assert env.stats()["agent_steps"] == A * N * (steps + 204) and env.stats()["obst_overflow"] == 0
    print("%s: overlapping pairs per arena-step %.2f in the first 600 steps, %.2f after %d" % (name, early, late, steps))
    env.close()


def test_processed_obstacle_table_equals_oracle():
    """ca_set_obstacles cuts edges like processObstacles; the vertex table (ids, links, coordinates) is the
    oracle's, so neighbour ids mean the same thing on both sides."""
    for scen, n in (("doorway", 10), ("congested", 12), ("blocks", 8), ("deadlock", 10), ("crowd", 16)):
        p = H.scenario_params(scen, n)
        g = H.make_gpu(1, n, scen, p)
        e = H.make_oracle(1, n, scen, p)
        tg, te = g.obstacle_table(), e.obstacle_table(cap=512)
        assert len(tg["next"]) == len(te["next"]) >= sum(len(q) for q in scenarios.obstacles(scen, n))
        np.testing.assert_array_equal(tg["verts"][:, 0], te["px"]); np.testing.assert_array_equal(tg["verts"][:, 1], te["py"])
        np.testing.assert_array_equal(tg["next"], te["next"]); np.testing.assert_array_equal(tg["convex"], te["convex"])
        g.close()


def test_run_to_run_determinism():
    """Two handles with the same configuration, seed and actions stay bit-identical (no dependence on wave timing,
    atomics order or which lane of a wave wins a merge): state, neighbour lists, observation, reward, counters."""
    from collision_avoidance_amd import _lib
    A, N = 256, 64
    p = scenarios.bench_params(N, 5.0, 10)
    g1 = H.make_gpu(A, N, "crowd", p, seed=21)
    g2 = H.make_gpu(A, N, "crowd", p, seed=21)
    rng = np.random.RandomState(5)
    fields = [_lib.FLD_POS_X, _lib.FLD_POS_Y, _lib.FLD_VEL_X, _lib.FLD_VEL_Y, _lib.FLD_GOAL_X, _lib.FLD_NB_COUNT, _lib.FLD_NB_IDX,
              _lib.FLD_OBST_COUNT, _lib.FLD_OBST_IDX, _lib.FLD_OBS, _lib.FLD_REWARD, _lib.FLD_REGOAL_COUNT]
    for s in range(120):
        act = rng.uniform(-0.5, 0.5, (A, N)).astype(np.float32)
        g1.step(act, stats=True); g2.step(act, stats=True)
        if s % 20 == 19:
            for f in fields:
                a, b = g1.get(f), g2.get(f)
                if f == _lib.FLD_NB_IDX:   # entries beyond the count are unspecified
                    cnt = g1.get(_lib.FLD_NB_COUNT)
                    mask = np.arange(a.shape[1])[None, :, None] < cnt[:, None, :]
                    a, b = np.where(mask, a, -1), np.where(mask, b, -1)
                if f == _lib.FLD_OBST_IDX:
                    cnt = g1.get(_lib.FLD_OBST_COUNT)
                    mask = np.arange(a.shape[1])[None, :, None] < cnt[:, None, :]
                    a, b = np.where(mask, a, -1), np.where(mask, b, -1)
                assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                      b.view(np.uint32) if b.dtype == np.float32 else b), (s, f)
    s1, s2 = g1.stats(), g2.stats()
    assert all(s1[k] == s2[k] for k in s1 if k != "sum_reward")
    g1.close(); g2.close()


def test_plain_c_client_of_the_abi(tmp_path):
    """tests/abi/c_client.c drives the library through include/ca_env.h from plain C (gcc, no Python, no HIP
    headers); its checksum of state, reward and observation equals the same run through the ctypes binding."""
    import shutil
    import subprocess
    from collision_avoidance_amd import _lib
    from collision_avoidance_amd import build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(B.LIB_PATH)
    exe = str(tmp_path / "c_client")
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    subprocess.check_call([gcc, "-std=c99", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi", "c_client.c"),
                           "-o", exe, "-L" + libdir, "-lcaenv", "-lm", "-Wl,-rpath," + libdir])
    A, N, steps = 8, 10, 40
    out = subprocess.run([exe, str(A), str(N), str(steps)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    c_hash, c_steps, c_coll = out.stdout.split()

    def fnv(h, a):
        for b in np.ascontiguousarray(a).view(np.uint8).reshape(-1).tolist():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    env = H.make_gpu(A, N, "doorway", H.scenario_params("doorway", N), seed=7, max_obst_neighbors=16)
    env.reset()
    lcg = 12345
    for s in range(steps):
        act = np.empty(A * N, np.float32)
        for q in range(A * N):
            lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
            act[q] = (np.float32(lcg >> 8) / np.float32(16777216.0) - np.float32(0.5)) * np.float32(1.5)
        env.step(act.reshape(A, N), stats=True)
    h = 1469598103934665603
    for f in (_lib.FLD_POS_X, _lib.FLD_POS_Y, _lib.FLD_REWARD, _lib.FLD_OBS):
        h = fnv(h, env.get(f))
    st = env.stats()
    assert (int(c_hash, 16), int(c_steps), int(c_coll)) == (h, st["agent_steps"], st["collisions"])
    env.close()


def test_world_without_obstacles_and_ragged_edges():
    """Empty obstacle table (no walls at all), a single agent per arena next to a wall, and an arena count that
    does not fill the last workgroup: the ragged and empty cases of the reference's own usage."""
    from collision_avoidance_amd import _lib
    p = H.scenario_params("crowd", 9)
    g = H.make_gpu(7, 9, "crowd", p, seed=2, polys=[], max_obst_neighbors=1)
    e = H.make_oracle(7, 9, "crowd", p, seed=2, polys=[], max_obst_neighbors=1)
    g.reset(); e.reset()
    rng = np.random.RandomState(0)
    for s in range(120):
        act = rng.uniform(-1, 1, (7, 9)).astype(np.float32)
        g.step(act, stats=True); e.step(act, flags=o.F_OBS | o.F_STATS)
    H.assert_state_equal(g, e, "no obstacles", obs=True, reward=True)
    H.assert_stats_equal(g, e, "no obstacles")
    assert g.obstacle_table()["verts"].shape == (0, 2) and (g.get(_lib.FLD_OBST_COUNT) == 0).all()
    g.close()
    p1 = H.scenario_params("crowd", 1)
    g = H.make_gpu(67, 1, "crowd", p1, seed=3)      # 67 arenas x 1 lane: two workgroups, the second nearly empty
    e = H.make_oracle(67, 1, "crowd", p1, seed=3)
    for s in range(200):
        g.orca_step(with_obs=True, stats=True); e.orca_step(flags=o.F_OBS | o.F_STATS)
    H.assert_state_equal(g, e, "single agents", obs=True)
    H.assert_stats_equal(g, e, "single agents")
    g.close()


def test_frozen_large_arenas_stay_untouched():
    """CA_F_FREEZE with the helper-lane neighbour scan (200 agents per arena): an arena whose arena_done flag is set
    is left exactly as it is -- state, lists, counters -- while its neighbours in the batch advance (ALAN:121-123)."""
    from collision_avoidance_amd import _lib
    A, N = 3, 200
    p = H.scenario_params("crowd", N)
    g = H.make_gpu(A, N, "crowd", p, seed=6)
    e = H.make_oracle(A, N, "crowd", p, seed=6)
    for env, step in ((g, lambda: g.orca_step(stats=True, freeze=True)), (e, lambda: e.orca_step(flags=o.F_STATS | o.F_FREEZE))):
        step(); step()
    H.assert_state_equal(g, e, "before freezing")
    done = np.array([0, 1, 0], np.int32)
    g.set(_lib.FLD_ARENA_DONE, done); e.set(o.FLD_ARENA_DONE, done)
    before = g.get(_lib.FLD_POS_X)[1].copy()
    for s in range(6):
        g.orca_step(stats=True, freeze=True); e.orca_step(flags=o.F_STATS | o.F_FREEZE)
    H.assert_state_equal(g, e, "one arena frozen")
    H.assert_stats_equal(g, e, "one arena frozen")
    np.testing.assert_array_equal(g.get(_lib.FLD_POS_X)[1], before)
    assert not np.array_equal(g.get(_lib.FLD_POS_X)[0], before)
    g.close()


def test_zero_copy_field_views():
    """VecCollisionAvoidanceEnv hands reward / done / any field out as torch views of the library's buffers."""
    import torch
    from collision_avoidance_amd import _lib
    env = H.make_gpu(12, 16, "crowd", scenarios.bench_params(16, 1.5, 5), use_torch=True)
    env.reset()
    act = torch.rand((12, 16), device="cuda") - 0.5
    for s in range(10):
        obs, rew, done, _ = env.step(act)
    torch.cuda.synchronize()
    assert rew.data_ptr() == env.field_tensor(_lib.FLD_REWARD).data_ptr() and rew.shape == (12, 16)
    np.testing.assert_array_equal(rew.cpu().numpy(), env.get(_lib.FLD_REWARD))
    np.testing.assert_array_equal(done.cpu().numpy(), env.get(_lib.FLD_ARENA_DONE))
    g = env.field_tensor(_lib.FLD_GOAL_X)
    assert g.dtype == torch.float64
    np.testing.assert_array_equal(g.cpu().numpy(), env.get(_lib.FLD_GOAL_X))
    assert obs.data_ptr() == env.field_tensor(_lib.FLD_OBS).data_ptr()      # the bound observation tensor
    env.close()


def test_host_results_in_the_environments_own_buffers():
    """numpy mode, step(copy=False): observation, reward and done flags land in the environment's own (page-locked) host
    buffers -- the same arrays every call, mutated in place, as the reference hands its dicts out (env.py:463-466) -- and hold
    what fresh copies hold; get(out=...) refuses a buffer of the wrong shape."""
    from collision_avoidance_amd import _lib
    env = H.make_gpu(12, 16, "crowd", scenarios.bench_params(16, 1.5, 5), use_torch=False)
    env.reset()
    rng = np.random.RandomState(3)
    first = None
    for s in range(6):
        obs, rew, done, _ = env.step(rng.uniform(-0.5, 0.5, (12, 16)).astype(np.float32), copy=False)
        if first is None:
            first = (obs, rew, done)
        assert obs is first[0] and rew is first[1] and done is first[2]
        np.testing.assert_array_equal(obs, env.get(_lib.FLD_OBS))
        np.testing.assert_array_equal(rew, env.get(_lib.FLD_REWARD))
        np.testing.assert_array_equal(done, env.get(_lib.FLD_ARENA_DONE))
    assert obs.shape == (12, 16, 64) and obs.dtype == np.float32 and np.abs(obs).max() > 0
    with pytest.raises(ValueError):
        env.get(_lib.FLD_OBS, out=np.empty((12, 16), np.float32))
    env.close()


@pytest.mark.parametrize("with_alan", [False, True], ids=["env", "alan"])
def test_checkpoint_and_resume_continue_bit_for_bit(with_alan, tmp_path):
    """get_state() / set_state(): an environment restored from a snapshot (through a file) continues exactly like the one
    that was never stopped -- state, lists, observation, reward, and the draws of re-goals, auto-resets and the ALAN bandit
    (counter-based, keyed by the restored counters).  The reference never serialises its env."""
    from collision_avoidance_amd import _lib, alan
    A, N = 40, 24
    p = scenarios.bench_params(N, 3.0, 10)
    p.update(max_step=17)                       # episodes end inside the run: auto-reset draws new starts
    if with_alan:
        p = H.scenario_params("crowd", N, max_step=60)
    a = H.make_gpu(A, N, "crowd", p, seed=9)
    if with_alan:
        a.alan_configure(alan.DEFAULT_ACTIONS)
    rng = np.random.RandomState(1)
    acts = rng.uniform(-0.7, 0.7, (45, A, N)).astype(np.float32)

    def advance(env, lo, hi):
        for s in range(lo, hi):
            if with_alan:
                env.alan_step(stats=True, freeze=True, with_obs=True)
            else:
                env.step(acts[s], stats=True, autoreset=True)
    advance(a, 0, 25)
    st = a.get_state()
    assert "REWARD" in st and (not with_alan or "ALAN_ACTION" in st)
    np.savez(tmp_path / "ckpt.npz", **st)
    advance(a, 25, 45)
    b = H.make_gpu(A, N, "crowd", p, seed=9)    # a fresh handle: its own scenario draws are overwritten by the snapshot
    if with_alan:
        b.alan_configure(alan.DEFAULT_ACTIONS)
    with np.load(tmp_path / "ckpt.npz") as z:
        b.set_state({k: z[k] for k in z.files})
    for f in [_lib.FLD_REWARD] + ([_lib.FLD_ALAN_ACTION] if with_alan else []):      # readable right after the restore
        assert np.array_equal(b.get(f), st["REWARD" if f == _lib.FLD_REWARD else "ALAN_ACTION"])
    a2 = H.make_gpu(A, N, "crowd", p, seed=9)     # ... and an observation taken right after the restore is the snapshot's own
    if with_alan:
        a2.alan_configure(alan.DEFAULT_ACTIONS)
    a2.set_state(st)
    assert np.array_equal(a2.observe(), b.observe())
    a2.close()
    advance(b, 25, 45)
    fields = [getattr(_lib, "FLD_" + n) for n in a._STATE_FIELDS] + [_lib.FLD_OBS, _lib.FLD_REWARD]
    if with_alan:
        fields += [_lib.FLD_ALAN_WEIGHTS, _lib.FLD_ALAN_TIMES, _lib.FLD_ALAN_ACTION]
    for f in fields:
        x, y = a.get(f), b.get(f)
        if f in (_lib.FLD_NB_IDX, _lib.FLD_OBST_IDX):    # entries beyond the count are unspecified
            cnt = a.get(_lib.FLD_NB_COUNT if f == _lib.FLD_NB_IDX else _lib.FLD_OBST_COUNT)
            m = np.arange(x.shape[1])[None, :, None] < cnt[:, None, :]
            x, y = np.where(m, x, -1), np.where(m, y, -1)
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), "field %d differs after the resume" % f
    assert a.get(_lib.FLD_EPISODE).max() > 0 or with_alan           # the run really crossed episode ends
    with pytest.raises(ValueError):
        H.make_gpu(A + 1, N, "crowd", p, seed=9).set_state(st)
    a.close(); b.close()


def test_step_packed_is_the_step_in_one_round_trip():
    """ca_step_packed: actions in, observation | reward | arena_done | step_count out in one copy -- the same bits as
    ca_step_host + three ca_get, for the full step, the ORCA-only step and with an externally bound observation buffer;
    the drop-in class (one environment, the reference's usage: run_rllib.py:77, 108) goes through it."""
    from collision_avoidance_amd import _lib
    A, N = 3, 10
    p = scenarios.env_params()
    a = H.make_gpu(A, N, "doorway", p, seed=4)
    b = H.make_gpu(A, N, "doorway", p, seed=4)
    rng = np.random.RandomState(2)
    for s in range(40):
        act = rng.uniform(-0.6, 0.6, (A, N)).astype(np.float32)
        if s % 7 == 6:
            ob_a = a.orca_step(with_obs=True, no_done=True)
            ob_b, rew_b, done_b, cnt_b = b.step_packed(None, no_done=True)
            rew_a, done_a = a.get(_lib.FLD_REWARD), a.get(_lib.FLD_ARENA_DONE)
        else:
            ob_a, rew_a, done_a, _ = a.step(act, stats=True, autoreset=(s % 5 == 0))
            ob_b, rew_b, done_b, cnt_b = b.step_packed(act, stats=True, autoreset=(s % 5 == 0))
        for x, y, what in ((ob_a, ob_b, "obs"), (rew_a, rew_b, "reward"), (done_a, done_b, "done"), (a.get(_lib.FLD_STEP_COUNT), cnt_b, "steps")):
            assert np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)), (s, what)
    orc = H.make_oracle(A, N, "doorway", p, seed=4)
    rng = np.random.RandomState(2)
    for s in range(40):
        act = rng.uniform(-0.6, 0.6, (A, N)).astype(np.float32)
        if s % 7 == 6:
            orc.orca_step(flags=o.F_OBS | o.F_NODONE)
        else:
            orc.step(act, flags=o.F_OBS | o.F_STATS | (o.F_AUTORESET if s % 5 == 0 else 0))
    H.assert_state_equal(b, orc, "packed steps", obs=True)
    with pytest.raises(RuntimeError, match="need"):
        b._call("ca_step_packed", b.h, None, 0, b._packed[0].ctypes.data, 8)
    with pytest.raises(ValueError):
        b.get(_lib.FLD_ALAN_WEIGHTS, out=np.zeros((A, 1, N)))
    a.close(); b.close()
    t = H.make_gpu(2, 6, "crowd", H.scenario_params("crowd", 6), seed=1, use_torch=True)     # observation bound to a torch tensor
    u = H.make_gpu(2, 6, "crowd", H.scenario_params("crowd", 6), seed=1)
    act = np.linspace(-0.5, 0.5, 12).astype(np.float32).reshape(2, 6)
    ob_t, rew_t, done_t, cnt_t = t.step_packed(act)
    ob_u, rew_u, done_u, _ = u.step(act)
    assert np.array_equal(ob_t, ob_u) and np.array_equal(rew_t, rew_u) and np.array_equal(done_t, done_u)
    t.close(); u.close()


def test_magnitudes_and_lds_are_checked_at_the_boundary():
    """obstacle tables outside the supported magnitudes are refused (the previous table stays), and a handle reads
    its diagnostic environment switches once, at ca_create."""
    import os
    from collision_avoidance_amd import _lib
    env = H.make_gpu(2, 8, "crowd", H.scenario_params("crowd", 8), seed=1)
    before = env.obstacle_table()["verts"].copy()
    for poly, what in (([(0, 0), (2e5, 0), (1, 1)], "beyond"), ([(0, 0), (1e-6, 0), (1, 1)], "shorter"),
                       ([(0, 0), (float("nan"), 0), (1, 1)], "beyond")):
        with pytest.raises(RuntimeError, match=what):
            env.set_obstacles([poly])
    np.testing.assert_array_equal(env.obstacle_table()["verts"], before)
    lanes = env.launch_info()["lanes_per_agent"]
    os.environ["CA_QUAD"] = "0"; os.environ["CA_REG_LINES"] = "0"
    try:
        env.set_obstacles(scenarios.obstacles("crowd", 8))           # re-selects the kernel: from the LATCHED switches
        assert env.launch_info()["lanes_per_agent"] == lanes
        env.step(np.zeros((2, 8), np.float32))
    finally:
        del os.environ["CA_QUAD"]; del os.environ["CA_REG_LINES"]
    env.close()
