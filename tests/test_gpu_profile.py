"""ca_profile / ca_profile_read (include/ca_env.h): the per-kernel times behind bench.py's `roofline` block.  A sampled
launch carries its start and stop event on its own dispatch; sampling must not change a result, must count what it was
asked to sample, and the times it reports must fit inside the wall time of the steps they belong to."""
import time

import numpy as np
import pytest

from collision_avoidance_amd import _lib, scenarios
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("A,N,K,nd", [(256, 64, 10, 5.0), (64, 16, 5, 1.5)])
def test_sampled_kernel_times_are_consistent_and_change_nothing(A, N, K, nd):
    p = scenarios.bench_params(N, nd, K)
    prof = H.make_gpu(A, N, "crowd", p, seed=4)
    plain = H.make_gpu(A, N, "crowd", p, seed=4)
    rng = np.random.RandomState(1)
    acts = rng.uniform(-0.5, 0.5, (8, A, N)).astype(np.float32)
    for s in range(40):                       # warm both up (first launches load code objects)
        prof.step(acts[s % 8], with_obs=True, stats=True)
        plain.step(acts[s % 8], with_obs=True, stats=True)
    prof.profile(2)
    prof.profile_read()
    prof.sync()
    steps = 64
    t0 = time.perf_counter()
    for s in range(steps):
        prof.step(acts[s % 8], with_obs=True, stats=True)
    prof.sync()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    for s in range(steps):
        plain.step(acts[s % 8], with_obs=True, stats=True)
    t = prof.profile_read()
    prof.profile(0)
    assert t["step_kernel"][0] == steps // 2 and t["obs_kernel"][0] == steps // 2, t
    assert t["nbr_kernel"][0] == 0                      # the neighbour search is fused into the solve launch
    ks, ko = t["step_kernel"][1], t["obs_kernel"][1]
    assert 1e-3 < ks < 5.0 and 1e-3 < ko < 5.0, t      # milliseconds, of the order of a launch
    assert ks + ko <= wall_ms * 1.05, (ks, ko, wall_ms)  # execution times fit inside the step they belong to
    for f in (_lib.FLD_POS_X, _lib.FLD_POS_Y, _lib.FLD_VEL_X, _lib.FLD_VEL_Y, _lib.FLD_OBS, _lib.FLD_REWARD):
        assert np.array_equal(prof.get(f), plain.get(f)), f
    assert prof.profile_read()["step_kernel"][0] == 0   # read clears, and nothing is sampled once switched off
    prof.close(); plain.close()


def test_small_kernels_are_timed_on_their_own_dispatch():
    """Kind 3 of ca_profile_read (reset_kernel, reset_arena_kernel, the ALAN select / update kernels): each launch carries
    its own start / stop event like the solve and observation kernels -- a sane count and a sane time after ca_reset and
    ca_alan_step, also with recycled events."""
    A, N = 64, 16
    p = H.scenario_params("crowd", N)
    import os
    os.environ["CA_ALAN_FUSED"] = "0"       # the three-launch form of the ALAN step (large batches take it; this small one would
    try:                                    # run the bandit inside the four-lanes kernel's launch)
        env = H.make_gpu(A, N, "crowd", p, seed=2)
        env.alan_configure([[1.0, 0.0], [0.7, 0.7], [0.7, -0.7], [0.0, 1.0]])
    finally:
        del os.environ["CA_ALAN_FUSED"]
    for rnd in range(2):                      # the second round reuses the events the first one handed back
        env.profile(1)
        env.profile_read()
        env.reset(with_obs=False)             # reset_kernel + reset_arena_kernel: 2 launches
        for s in range(5):
            env.alan_step(stats=True)         # select + solve + update: 2 small launches and 1 solve launch each
        t = env.profile_read()
        env.profile(0)
        assert t["reset_kernels"][0] == 2 + 2 * 5, t
        assert t["step_kernel"][0] == 5 and t["obs_kernel"][0] == 0, t
        assert 1e-4 < t["reset_kernels"][1] < 1.0, t          # milliseconds: a few microseconds each, never negative / stale
        assert 1e-3 < t["step_kernel"][1] < 5.0, t
    env.close()


def test_one_launch_rollout_is_reported_per_step():
    """ca_rollout's one-launch form advances T steps per launch (at most 256): ca_profile_read reports such a launch per
    step, so a consumer never sees a `step_kernel` time that is T times a step's."""
    A, N = 64, 16
    p = scenarios.bench_params(N, 1.5, 5)
    env = H.make_gpu(A, N, "crowd", p, seed=2)
    assert env.launch_info()["rollout_one_launch"] == 1
    env.rollout(50, stats=True)
    env.profile(1)
    env.profile_read()
    for s in range(20):
        env.orca_step(stats=True)
    single = env.profile_read()["step_kernel"]
    env.rollout(600, stats=True)              # 256 + 256 + 88 steps: three launches
    multi = env.profile_read()["step_kernel"]
    env.profile(0)
    assert single[0] == 20 and multi[0] == 3, (single, multi)
    assert multi[1] < 1.5 * single[1], (single, multi)      # per step (one launch per step pays the launch on top)
    assert env.stats()["agent_steps"] == A * N * (50 + 20 + 600)
    env.close()
