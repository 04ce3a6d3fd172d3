"""Stream ordering of the library (round-1 failure: fills issued on the null stream raced the handle's own
non-blocking stream, so the tail of a 36-byte mask could be zeroed after it had been uploaded).

Every case builds FRESH handles that run on their own stream (use_torch=False) with sizes that are not
multiples of 8 / 16 bytes and makes the hazardous call the very first one: reset_masked, alan_step right
after alan_configure, reset + step right after init_scenario.  All results are compared with the oracle
(env.py:461-488 reset semantics; ALAN_true.py:569-628).  Also: tensors that do not live on the handle's
device take the host path instead of being dereferenced as device memory."""
import numpy as np
import pytest

from collision_avoidance_amd import _lib, alan
from oracle import oracle as o
from tests import helpers as H

pytestmark = pytest.mark.gpu

SIZES = [(1, 5), (3, 7), (9, 12), (67, 3)]


def test_fresh_handle_first_call_reset_masked():
    rng = np.random.RandomState(11)
    for rep in range(25):
        for A, N in SIZES:
            p = H.scenario_params("doorway", N, max_step=30)
            g = H.make_gpu(A, N, "doorway", p, seed=rep)
            e = H.make_oracle(A, N, "doorway", p, seed=rep)
            mask = (rng.uniform(size=A) < 0.6).astype(np.int32)
            mask[-1] = 1                                     # the entry a late tail fill would wipe
            g.reset_masked(mask); e.reset_masked(mask)
            H.assert_state_equal(g, e, "first-call masked reset A=%d rep=%d" % (A, rep), obs=True)
            g.close()


def test_fresh_handle_first_call_reset_and_step():
    rng = np.random.RandomState(12)
    for rep in range(25):
        for A, N in SIZES:
            p = H.scenario_params("crowd", N)
            g = H.make_gpu(A, N, "crowd", p, seed=100 + rep)
            e = H.make_oracle(A, N, "crowd", p, seed=100 + rep)
            g.reset(); e.reset()
            act = rng.uniform(-1, 1, (A, N)).astype(np.float32)
            g.step(act, stats=True); e.step(act, flags=o.F_OBS | o.F_STATS)
            H.assert_state_equal(g, e, "first-call reset+step A=%d rep=%d" % (A, rep), obs=True, reward=True)
            H.assert_stats_equal(g, e)
            g.close()


def test_fresh_handle_first_call_alan_step():
    for rep in range(25):
        for A, N in SIZES:
            p = H.scenario_params("crowd", N)
            g = H.make_gpu(A, N, "crowd", p, seed=200 + rep)
            e = H.make_oracle(A, N, "crowd", p, seed=200 + rep)
            acts = alan.DEFAULT_ACTIONS[:3 + rep % 5]
            g.alan_configure(acts); e.alan_configure(acts)
            for _ in range(3):
                g.alan_step(); e.alan_step()
            H.assert_state_equal(g, e, "first-call alan_step A=%d rep=%d" % (A, rep))
            H._eq(g.get(_lib.FLD_ALAN_ACTION), e.get(o.FLD_ALAN_ACTION), "alan action")
            H._eq(g.get(_lib.FLD_ALAN_WEIGHTS), e.get(o.FLD_ALAN_WEIGHTS), "alan weights")
            H._eq(g.get(_lib.FLD_ALAN_TIMES), e.get(o.FLD_ALAN_TIMES), "alan times")
            g.close()


def test_fresh_handle_without_obstacles_first_call_step():
    """ca_create -> ca_set_obstacles(n_poly = 0) -> first step: no call in between synchronised in round 1."""
    for rep in range(20):
        A, N = 5, 9
        p = H.scenario_params("crowd", N)
        g = H.make_gpu(A, N, "crowd", p, seed=300 + rep, polys=[])
        e = H.make_oracle(A, N, "crowd", p, seed=300 + rep, polys=[])
        g.orca_step(with_obs=True, stats=True); e.orca_step(flags=o.F_OBS | o.F_STATS)
        H.assert_state_equal(g, e, "no-obstacle first step rep=%d" % rep, obs=True)
        g.close()


@pytest.mark.parametrize("use_torch", [False, True])
def test_fresh_handle_tensors_off_device_take_the_host_path(use_torch):
    torch = pytest.importorskip("torch")
    A, N = 9, 12
    p = H.scenario_params("doorway", N, max_step=30)
    g = H.make_gpu(A, N, "doorway", p, seed=4, use_torch=use_torch)
    e = H.make_oracle(A, N, "doorway", p, seed=4)
    mask = np.array([1, 0, 0, 1, 0, 1, 0, 0, 1], np.int32)
    g.reset_masked(torch.tensor(mask)); e.reset_masked(mask)                    # CPU tensor
    H.assert_state_equal(g, e, "cpu-tensor mask", obs=True)
    m2 = 1 - mask
    g.reset_masked(torch.tensor(m2, device="cuda")); e.reset_masked(m2)        # device tensor
    H.assert_state_equal(g, e, "device-tensor mask", obs=True)
    rng = np.random.RandomState(0)
    px = rng.uniform(5, 10, (A, N)).astype(np.float32); py = rng.uniform(0, 10, (A, N)).astype(np.float32)
    g.reset(torch.tensor(px), torch.tensor(py)); e.reset(px, py)                # CPU tensors
    H.assert_state_equal(g, e, "cpu-tensor reset", obs=True)
    px2 = px[::-1].copy()
    g.reset(torch.tensor(px2, device="cuda"), torch.tensor(py, device="cuda")); e.reset(px2, py)
    H.assert_state_equal(g, e, "device-tensor reset", obs=True)
    g.reset(torch.tensor(px, device="cuda"), torch.tensor(py)); e.reset(px, py)  # mixed: host path
    H.assert_state_equal(g, e, "mixed-tensor reset", obs=True)
    acts = alan.DEFAULT_ACTIONS
    g.alan_configure(acts); e.alan_configure(acts)
    u = rng.uniform(size=(A, N))
    g.alan_step(torch.tensor(u)); e.alan_step(u)                                # CPU tensor
    u2 = rng.uniform(size=(A, N))
    g.alan_step(torch.tensor(u2, device="cuda")); e.alan_step(u2)               # device tensor
    H.assert_state_equal(g, e, "tensor uniforms")
    H._eq(g.get(_lib.FLD_ALAN_ACTION), e.get(o.FLD_ALAN_ACTION), "alan action")
    with pytest.raises(ValueError):
        g.reset_masked(torch.zeros(A + 1, dtype=torch.int32, device="cuda"))
    g.close()
