"""The trainer-facing adapters on the real HIP environment: identical outputs to the same adapter on the
oracle-backed stand-in (bit-exact observations/rewards, same done / episode / collision bookkeeping),
masked reset, per-arena statistics."""
import numpy as np
import pytest

from collision_avoidance_amd import _lib, adapters
from oracle import oracle as o
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_masked_reset_and_arena_stats_match_oracle():
    A, N = 9, 12
    p = H.scenario_params("doorway", N, max_step=30)
    g = H.make_gpu(A, N, "doorway", p, seed=6)
    e = H.make_oracle(A, N, "doorway", p, seed=6)
    g.reset(); e.reset()
    rng = np.random.RandomState(1)
    for s in range(70):
        act = rng.uniform(-1, 1, (A, N)).astype(np.float32)
        g.step(act, stats=True); e.step(act, flags=o.F_OBS | o.F_STATS)
        if s in (10, 29, 45):
            mask = (rng.uniform(size=A) < 0.4).astype(np.int32)
            if s == 29:
                mask = e.get(o.FLD_ARENA_DONE).copy()        # exactly the arenas that just hit the cap
                assert mask.any() and not mask.all()     # those reset at step 10 are mid-episode
                mask[::2] = 0
            g.reset_masked(mask); e.reset_masked(mask)
            H.assert_state_equal(g, e, "masked reset at %d" % s, obs=True)
    H.assert_state_equal(g, e, "end", obs=True, reward=True)
    gs, es = g.get(_lib.FLD_ARENA_STATS), e.get(o.FLD_ARENA_STATS)
    np.testing.assert_array_equal(gs[:, [0, 1, 2, 3, 4, 6, 7]], es[:, [0, 1, 2, 3, 4, 6, 7]])
    np.testing.assert_allclose(gs[:, 5].copy().view(np.float64), es[:, 5].copy().view(np.float64), rtol=1e-9, atol=1e-9)
    st = g.arena_stats()
    assert (st["last_episode_steps"][st["episodes"] > 0] >= 30).all()    # without auto-reset an arena keeps stepping past its cap
    g.close()


@pytest.mark.parametrize("use_torch", [False, True])
def test_agent_vector_env_equals_oracle_backed(use_torch):
    A, N = 16, 10
    p = H.scenario_params("doorway", N, max_step=60)
    vg = adapters.AgentVectorEnv(H.make_gpu(A, N, "doorway", p, seed=8, use_torch=use_torch), new_step_api=True)
    ve = adapters.AgentVectorEnv(H.OracleVec(A, N, "doorway", p, seed=8), new_step_api=True)
    og, _ = vg.reset(); oe, _ = ve.reset()
    host = (lambda x: x.cpu().numpy()) if use_torch else np.asarray
    np.testing.assert_array_equal(host(og), oe)
    rng = np.random.RandomState(3)
    ends = 0
    for s in range(150):
        act = rng.uniform(-0.7, 0.7, (A * N, 1)).astype(np.float32)
        if use_torch:
            import torch
            rg = vg.step(torch.as_tensor(act, device="cuda"))
        else:
            rg = vg.step(act)
        re = ve.step(act)
        for k in range(4):
            H._eq(host(rg[k]), re[k], "step %d output %d" % (s, k))
        for k in ("collisions", "obst_collisions", "goals_reached", "agent_arrived"):
            np.testing.assert_array_equal(rg[4][k], re[4][k])
        for k in ("arena", "length", "arrived", "truncated"):
            np.testing.assert_array_equal(rg[4]["episode"][k], re[4]["episode"][k])
        ends += len(rg[4]["episode"]["arena"])
    assert ends >= A
    vg.close()


def test_multi_agent_vector_env_equals_oracle_backed():
    A, N = 4, 6
    p = H.scenario_params("crowd", N, max_step=300)
    mg = adapters.MultiAgentVectorEnv(H.make_gpu(A, N, "crowd", p, seed=5), per_agent_dones=True)
    me = adapters.MultiAgentVectorEnv(H.OracleVec(A, N, "crowd", p, seed=5), per_agent_dones=True)
    og, oe = mg.vector_reset(), me.vector_reset()
    rng = np.random.RandomState(2)
    resets = 0
    for s in range(350):
        assert [sorted(d) for d in og] == [sorted(d) for d in oe]
        acts = [{aid: [float(rng.uniform(-0.3, 0.3))] for aid in d} for d in og]
        og, rg, dg, ig = mg.vector_step(acts)
        oe, re, de, ie = me.vector_step(acts)
        assert dg == de and rg == re and ig == ie
        for e in range(A):
            for aid in og[e]:
                H._eq(og[e][aid], oe[e][aid], "obs %d %d %s" % (s, e, aid))
            if dg[e]['__all__']:
                og[e], oe[e] = mg.reset_at(e), me.reset_at(e)
                resets += 1
    assert resets >= A
    mg.close()
