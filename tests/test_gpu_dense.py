"""Arenas whose agent count is no power of two, packed back to back in a wave (the lane kernels: N lanes per arena instead of
the next power of two; the observation: 16 consecutive agents of the batch per workgroup) -- the reference env's own shape
is 10 agents per environment (env.py:26).  Forced onto the lane kernel (small batches would take the four-lanes kernel),
against the oracle bit for bit, and against the same handle with the packing switched off."""
import os

import numpy as np
import pytest

from collision_avoidance_amd import _lib
from oracle import oracle as o
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _make(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("scenario,N,A", [("doorway", 10, 77), ("crowd", 3, 50), ("crowd", 12, 31), ("deadlock", 20, 13),
                                          ("incoming", 17, 10), ("blocks", 6, 41), ("crowd", 33, 9)])
def test_dense_packing_full_steps_bit_exact(scenario, N, A):
    p = H.scenario_params(scenario, N, max_step=25)
    g = _make({"CA_QUAD": "0"}, lambda: H.make_gpu(A, N, scenario, p, seed=11))
    ref = _make({"CA_QUAD": "0", "CA_DENSE": "0", "CA_OBS_DENSE": "0"}, lambda: H.make_gpu(A, N, scenario, p, seed=11))
    e = H.make_oracle(A, N, scenario, p, seed=11)
    li, lr = g.launch_info(), ref.launch_info()
    assert li["lanes_per_agent"] == 1 and lr["lanes_per_agent"] == 1
    P = 1
    while P < N:
        P *= 2
    if 64 // N > 64 // P:
        assert li["grid"] == -(-A // (64 // N)) and lr["grid"] == -(-A // (64 // P)), (li, lr)   # more arenas per wave
    if N < 16:
        assert li["obs_grid"] == -(-A * N // 16) and lr["obs_grid"] == A, (li, lr)
    g.reset(); ref.reset(); e.reset()
    rng = np.random.RandomState(4)
    for s in range(60):
        act = rng.uniform(-1.0, 1.0, (A, N)).astype(np.float32)
        g.step(act, stats=True, autoreset=True)
        ref.step(act, stats=True, autoreset=True)
        e.step(act, flags=o.F_OBS | o.F_STATS | o.F_AUTORESET)
        if s % 15 == 14:
            H.assert_state_equal(g, e, "%s dense step %d" % (scenario, s), obs=True, reward=True)
            H.assert_state_equal(ref, e, "%s padded step %d" % (scenario, s), obs=True, reward=True)
    H.assert_stats_equal(g, e, scenario)
    np.testing.assert_array_equal(g.get(_lib.FLD_ARENA_STATS)[:, [0, 1, 2, 3, 4, 6, 7]], e.get(o.FLD_ARENA_STATS)[:, [0, 1, 2, 3, 4, 6, 7]])
    assert g.stats()["episodes"] >= A      # the cap of 25 steps ended every arena at least once (auto-reset inside the call)
    g.close(); ref.close()


def test_dense_packing_orca_rollout_freeze_and_large_batch():
    """The reference env's configuration as a large batch (4099 environments of 10 agents: the last wave is ragged), ORCA-only
    with per-arena freezing, all arenas against the oracle."""
    A, N = 4099, 10
    p = H.scenario_params("doorway", N, max_step=40)
    g = _make({"CA_QUAD": "0"}, lambda: H.make_gpu(A, N, "doorway", p, seed=2))
    e = H.make_oracle(A, N, "doorway", p, seed=2)
    assert g.launch_info()["lanes_per_agent"] == 1 and g.launch_info()["grid"] == -(-A // 6)
    sc = (np.arange(A) % 37).astype(np.int32)          # arenas hit the cap at different steps
    g.set(_lib.FLD_STEP_COUNT, sc); e.set(o.FLD_STEP_COUNT, sc)
    for s in range(30):
        g.orca_step(stats=True, freeze=True, with_obs=(s % 10 == 9))
    e.rollout(30, flags=o.F_STATS | o.F_FREEZE, n_threads=8)
    H.assert_state_equal(g, e, "doorway batch")
    H.assert_stats_equal(g, e, "doorway batch")
    assert 0 < int(g.get(_lib.FLD_ARENA_DONE).sum()) < A
    g.close()
