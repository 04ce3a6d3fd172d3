"""The four-lanes-per-agent solve kernel (csrc/ca_quad.h) against the oracle, bit for bit, with the variant FORCED
on (CA_QUAD=1; the library otherwise picks it only for batches that leave the chip short of waves), and its
T-steps-per-launch rollout against T single steps.  Reference semantics: collision_avoidence_env.py:385 (one doStep
per step), :447-458 / :570-573 and ALAN_true.py:106-123 (the ORCA-only loops ca_rollout replaces)."""
import os

import numpy as np
import pytest

from collision_avoidance_amd import scenarios
from oracle import oracle as o
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def forced_quad():
    old = os.environ.get("CA_QUAD")
    os.environ["CA_QUAD"] = "1"
    yield
    if old is None:
        del os.environ["CA_QUAD"]
    else:
        os.environ["CA_QUAD"] = old


def _is_quad(gpu):
    return gpu.launch_info()["lanes_per_agent"] == 4


CASES = [  # name, scenario, A, N, overrides, steps
    ("c2like", "crowd", 37, 16, dict(neighbor_dist=1.5, max_neighbors=5), 300),
    ("c3like", "crowd", 9, 64, dict(neighbor_dist=5.0, max_neighbors=10), 150),
    ("n128", "crowd", 3, 128, dict(neighbor_dist=5.0, max_neighbors=10), 60),
    ("circle8", "circle", 5, 8, {}, 400),
    ("circle33", "circle", 3, 33, {}, 200),
    ("tiny1", "crowd", 70, 1, dict(max_neighbors=5), 50),
    ("tiny3", "crowd", 41, 3, dict(max_neighbors=3), 200),
    ("separated", "crowd_separated", 5, 24, dict(neighbor_dist=3.0, max_neighbors=10), 300),
    ("free", "crowd", 5, 20, dict(neighbor_dist=2.0, max_neighbors=5), 200),
    ("k0", "crowd", 6, 12, dict(max_neighbors=0), 100),
    ("k7", "crowd", 6, 40, dict(max_neighbors=7, neighbor_dist=4.0), 150),
    # the reference's own worlds: more than four obstacle edges in range (the 16-neighbour instantiation, lines in rounds)
    ("doorway", "doorway", 40, 10, {}, 400),
    ("deadlock", "deadlock", 6, 30, {}, 500),
    ("blocks", "blocks", 9, 12, {}, 300),
    ("congested50", "congested", 3, 50, {}, 200),     # 256-lane workgroups, > 48 KB of LDS
    ("incoming", "incoming", 4, 17, {}, 200),
]


@pytest.mark.parametrize("name,scenario,A,N,over,steps", CASES, ids=[c[0] for c in CASES])
def test_quad_orca_rollout_bit_exact(name, scenario, A, N, over, steps):
    p = H.scenario_params(scenario, N, **over)
    polys = [] if name == "free" else None      # "free": no obstacles at all
    gpu = H.make_gpu(A, N, scenario, p, seed=11, polys=polys)
    assert _is_quad(gpu), gpu.launch_info()
    orc = H.make_oracle(A, N, scenario, p, seed=11, polys=polys)
    done = 0
    for chunk in (1, 2, 7, steps):          # 1 step, then launches of several steps each
        chunk = min(chunk, steps - done)
        if chunk <= 0:
            break
        gpu.rollout(chunk, stats=True)
        orc.rollout(chunk, flags=o.F_STATS)
        done += chunk
        H.assert_state_equal(gpu, orc, "%s after %d steps" % (name, done))
    H.assert_stats_equal(gpu, orc, name)
    gpu.close()


@pytest.mark.parametrize("N,K", [(16, 5), (64, 10), (5, 2)])
def test_quad_step_with_actions_obs_and_autoreset(N, K):
    A = 12
    p = scenarios.bench_params(N, 3.0, K)
    p.update(max_step=40, done_mode=1)
    gpu = H.make_gpu(A, N, "crowd", p, seed=5)
    assert _is_quad(gpu)
    orc = H.make_oracle(A, N, "crowd", p, seed=5)
    rng = np.random.RandomState(5)
    for s in range(130):   # crosses three episode ends (auto-reset inside the call)
        act = rng.uniform(-1.0, 1.0, (A, N)).astype(np.float32)
        gpu.step(act, stats=True, autoreset=True)
        orc.step(act, flags=o.F_OBS | o.F_STATS | o.F_AUTORESET)
        if s % 13 == 0 or s > 120:
            H.assert_state_equal(gpu, orc, "step %d" % s, obs=True, reward=True)
    H.assert_stats_equal(gpu, orc, "actions")
    gpu.close()


def test_quad_with_a_world_per_arena():
    """Per-arena obstacle tables (ca_set_obstacles_per_arena: ids local to each arena's table) under the quad kernel: a
    different triangle or wedge in every arena (at most four edges in range), several arenas per wave."""
    A, N = 22, 6
    p = H.scenario_params("crowd", N, neighbor_dist=3.0, max_neighbors=5)
    rng = np.random.RandomState(3)
    worlds = []
    for a in range(A):
        c = rng.uniform(2.0, 4.0, 2)
        if a % 3 == 2:
            worlds.append([])                                            # an arena without any obstacle
        elif a % 3 == 1:
            worlds.append([[(c[0], c[1]), (c[0] + 1.0, c[1] + 0.2), (c[0] + 0.3, c[1] + 1.1)]])          # counter-clockwise triangle
        else:
            worlds.append([[(c[0], c[1]), (c[0] + 0.2, c[1] + 1.0), (c[0] + 1.2, c[1] + 0.9), (c[0] + 1.0, c[1] - 0.1)][::-1]])
    g = H.make_gpu(A, N, "crowd", p, seed=8, polys=dict(per_arena=worlds), max_obst_neighbors=4)
    assert _is_quad(g)
    e = H.make_oracle(A, N, "crowd", p, seed=8, polys=dict(per_arena=worlds), max_obst_neighbors=4)
    rng = np.random.RandomState(8)
    for s in range(150):
        if s % 3 == 0:
            g.rollout(2, stats=True); e.rollout(2, flags=o.F_STATS)
        else:
            act = rng.uniform(-0.9, 0.9, (A, N)).astype(np.float32)
            g.step(act, stats=True); e.step(act, flags=o.F_OBS | o.F_STATS)
        if s % 25 == 0 or s > 140:
            H.assert_state_equal(g, e, "per-arena worlds, step %d" % s, obs=(s % 3 != 0))
    H.assert_stats_equal(g, e, "per-arena worlds")
    assert g.stats()["obst_collisions"] > 0 or g.get(5).any()       # the obstacles were in somebody's way
    g.close()


@pytest.mark.parametrize("done_mode", [0, 1, 2])
def test_quad_done_modes(done_mode):
    """The three done tests (env.py:352-365 x-threshold, ALAN:547-566 goal, the synthetic re-goal) in the quad kernel's
    T-steps-per-launch loop, with the step cap and auto-reset."""
    A, N = 9, 14
    p = scenarios.bench_params(N, 2.5, 5)
    p.update(done_mode=done_mode, max_step=90, done_x_thresh=3.0)
    g = H.make_gpu(A, N, "crowd", p, seed=4)
    assert _is_quad(g)
    e = H.make_oracle(A, N, "crowd", p, seed=4)
    for chunk in (1, 30, 45, 100, 13):
        g.rollout(chunk, stats=True, autoreset=True)
        e.rollout(chunk, flags=o.F_STATS | o.F_AUTORESET)
        H.assert_state_equal(g, e, "done_mode %d after a launch of %d" % (done_mode, chunk))
    H.assert_stats_equal(g, e, "done modes")
    assert g.stats()["episodes"] >= A
    g.close()


def test_quad_freeze_rollout_equals_single_steps_and_oracle():
    """Episodes of different lengths end each where the serial loop would (ALAN:121-123), inside ONE launch."""
    A, N = 10, 12
    p = H.scenario_params("circle", N)
    p.update(max_step=3000)
    one = H.make_gpu(A, N, "circle", p, seed=2)
    many = H.make_gpu(A, N, "circle", p, seed=2)
    orc = H.make_oracle(A, N, "circle", p, seed=2)
    # different arenas get different head starts so that they finish at different steps
    px, py = one.get(0), one.get(1)
    px[1::2] *= 1.0 + 0.01 * np.arange(px[1::2].shape[0])[:, None]
    for e in (one, many):
        e.set(0, px); e.set(1, py)
    orc.set(o.FLD_POS_X, px); orc.set(o.FLD_POS_Y, py)
    many.rollout(1200, stats=True, freeze=True)                      # one launch
    for _ in range(1200):
        one.orca_step(stats=True, freeze=True)                        # 1200 launches
    orc.rollout(1200, flags=o.F_STATS | o.F_FREEZE)
    H.assert_state_equal(many, orc, "freeze rollout vs oracle")
    H.assert_state_equal(one, orc, "freeze single steps vs oracle")
    H.assert_stats_equal(many, orc, "freeze")
    from collision_avoidance_amd import _lib
    np.testing.assert_array_equal(many.get(_lib.FLD_ARRIVE_STEP), one.get(_lib.FLD_ARRIVE_STEP))
    assert len(set(many.get(_lib.FLD_STEP_COUNT).tolist())) > 1      # the episodes really differ in length
    for e in (one, many):
        e.close()


def test_lane_steps_and_quad_rollouts_share_one_handle():
    """At exactly one lane-wave per SIMD the handle steps with one lane per agent and rolls out with four (the two
    kernels work on the same state arrays): alternate them and compare with the oracle."""
    del os.environ["CA_QUAD"]
    try:
        A, N = 4096, 16
        p = scenarios.bench_params(N, 1.5, 5)
        gpu = H.make_gpu(A, N, "crowd", p, seed=13)
        info = gpu.launch_info()
        assert info["lanes_per_agent"] == 1 and info["rollout_one_launch"] == 1, info
        orc = H.make_oracle(A, N, "crowd", p, seed=13)
        rng = np.random.RandomState(13)
        for block in range(3):
            for s in range(2):
                act = rng.uniform(-0.7, 0.7, (A, N)).astype(np.float32)
                gpu.step(act, stats=True)
                orc.step_mt(act, flags=o.F_OBS | o.F_STATS, n_threads=8)
            H.assert_state_equal(gpu, orc, "after steps, block %d" % block, obs=True, reward=True)
            gpu.rollout(6 + block, stats=True)
            orc.rollout(6 + block, flags=o.F_STATS, n_threads=8)
            H.assert_state_equal(gpu, orc, "after rollout, block %d" % block)
        H.assert_stats_equal(gpu, orc, "mixed")
        gpu.close()
    finally:
        os.environ["CA_QUAD"] = "1"


def test_quad_is_the_default_for_small_batches_only():
    del os.environ["CA_QUAD"]
    small = H.make_gpu(64, 16, "crowd", scenarios.bench_params(16, 1.5, 5))
    big = H.make_gpu(4096, 64, "crowd", scenarios.bench_params(64, 5.0, 10))
    assert _is_quad(small) and not _is_quad(big)
    small.close(); big.close()
    os.environ["CA_QUAD"] = "1"
