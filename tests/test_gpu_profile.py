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
