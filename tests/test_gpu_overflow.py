"""RVO2 keeps every obstacle edge in range of an agent (collision_avoidence_env.py:249, 301-318 iterate them all); the lists
of this library hold max_obst_neighbors <= 16.  A world that overflows them must not deviate silently: the step calls fail
with CA_ERANGE naming arena and agent, unless the caller accepted the truncation -- and then the result is the oracle's with
the same capacity, bit for bit, on every solve kernel."""
import math
import os

import numpy as np
import pytest

from tests import helpers as H
from oracle import oracle as o

pytestmark = pytest.mark.gpu


def ring_world(n_edges=24, radius=1.2, centre=(5.0, 5.0)):
    """A clockwise polygon (edges visible from inside, like the boundary walls of env.py:118) of `n_edges` short edges around
    `centre`: an agent near the centre has every one of them within timeHorizonObst * maxSpeed + radius = 2."""
    th = [-(2 * math.pi * k) / n_edges for k in range(n_edges)]       # clockwise
    return [np.array([[centre[0] + radius * math.cos(t), centre[1] + radius * math.sin(t)] for t in th], np.float32)]


def _setup(kind, allow, A=3, N=2, offset=40):
    p = H.scenario_params("doorway", N)
    env_over = {"quad": {"CA_QUAD": "1"}, "lane": {"CA_QUAD": "0"}, "table": {"CA_QUAD": "0", "CA_REG_LINES": "0"}}[kind]
    old = {k: os.environ.get(k) for k in env_over}
    os.environ.update(env_over)
    try:
        g = H.make_gpu(A, N, "doorway", p, seed=3, arena_offset=offset, max_obst_neighbors=16, polys=ring_world(),
                       allow_obst_overflow=allow)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    c = H.make_oracle(A, N, "doorway", p, seed=3, arena_offset=offset, max_obst_neighbors=16, polys=ring_world())
    rng = np.random.RandomState(5)
    px = (5.0 + rng.uniform(-0.3, 0.3, (A, N))).astype(np.float32)
    py = (5.0 + rng.uniform(-0.3, 0.3, (A, N))).astype(np.float32)
    px[1, :] = 40.0                                          # arena 1 stands far from the ring: no edge in range there
    for e, fx, fy in ((g, H._lib_fld("POS_X"), H._lib_fld("POS_Y")), (c, o.FLD_POS_X, o.FLD_POS_Y)):
        e.set(fx, px); e.set(fy, py)
    return g, c


@pytest.mark.parametrize("kind", ["quad", "lane", "table"])
def test_overflow_is_an_error_unless_accepted(kind):
    g, c = _setup(kind, allow=False)
    g.orca_step(stats=True)                                  # asynchronous: the launch itself succeeds ...
    with pytest.raises(RuntimeError) as ei:                  # ... and the synchronisation reports what the kernel met
        g.sync()
    msg = str(ei.value)
    assert "(-5)" in msg and "overflowed" in msg and "max_obst_neighbors=16" in msg and "collision_avoidence_env.py:249" in msg
    assert ("arena 40," in msg or "arena 42," in msg) and "24 obstacle edges" in msg, msg       # global arena id, edges in range
    for call in (lambda: g.orca_step(), lambda: g.step(np.zeros((3, 2), np.float32)), lambda: g.rollout(3),
                 lambda: g.step_packed(None)):
        with pytest.raises(RuntimeError, match="overflowed"):                                    # sticky
            call()
    st = g.stats()                                           # reading still works, and counts: 2 arenas x 2 agents x 1 step
    assert st["obst_overflow"] == 4 and st["agent_steps"] == 3 * 2
    assert g.arena_stats()["obst_overflow"].tolist() == [2, 0, 2]
    assert g.get(H._lib_fld("OBST_COUNT")).max() == 16
    g.reset_stats()                                          # clears the status; the next overflow raises again
    g.orca_step(stats=True)
    with pytest.raises(RuntimeError, match="overflowed"):
        g.sync()
    g.L.ca_allow_obstacle_overflow(g.h, 1)                   # accepted from here on
    g.orca_step(stats=True); g.sync()
    g.close()


@pytest.mark.parametrize("kind", ["quad", "lane", "table"])
def test_accepted_overflow_equals_the_oracle_with_the_same_capacity(kind):
    g, c = _setup(kind, allow=True)
    rng = np.random.RandomState(9)
    for s in range(40):
        act = rng.uniform(-1, 1, (3, 2)).astype(np.float32)
        g.step(act, stats=True); c.step(act, flags=o.F_OBS | o.F_STATS)
    g.sync()
    H.assert_state_equal(g, c, "ring/" + kind, obs=True, reward=True)
    H.assert_stats_equal(g, c, "ring/" + kind)
    assert g.stats()["obst_overflow"] > 0
    g.close()


def test_reference_sized_worlds_never_trip_it():
    """The reference's own doorway world (14 edges after processObstacles) with the default capacity: 300 steps, no status."""
    p = H.scenario_params("doorway", 10)
    g = H.make_gpu(4, 10, "doorway", p, seed=1)
    g.rollout(300, stats=True)
    g.sync()
    assert g.stats()["obst_overflow"] == 0
    g.close()
