"""Host logic of the trainer-facing adapters (collision_avoidance_amd/adapters.py) on top of the
oracle-backed stand-in for the device environment (tests/helpers.py::OracleVec)."""
import numpy as np
import pytest

from collision_avoidance_amd import _lib, adapters
from tests import helpers as H


def _params(N, max_step=40):
    return H.scenario_params("doorway", N, max_step=max_step)


def test_agent_vector_env_autoreset_convention():
    A, N = 5, 6
    env = adapters.AgentVectorEnv(H.OracleVec(A, N, "doorway", _params(N, 25), seed=4))
    assert env.num_envs == A * N and env.single_observation_space.shape == (64,) and env.action_space.shape == (A * N, 1)
    obs = env.reset()
    assert obs.shape == (A * N, 64)
    rng = np.random.RandomState(0)
    ends = 0
    for s in range(60):
        obs, rew, done, info = env.step(rng.uniform(-0.5, 0.5, (A * N, 1)).astype(np.float32))
        assert obs.shape == (A * N, 64) and rew.shape == (A * N,) and done.shape == (A * N,) and done.dtype == bool
        d = done.reshape(A, N)
        assert (d == d[:, :1]).all()                         # an arena's agents finish together
        ep = info["episode"]
        np.testing.assert_array_equal(ep["arena"], np.nonzero(d[:, 0])[0])
        if len(ep["arena"]):
            ends += len(ep["arena"])
            assert (ep["length"] == 25).all() and (ep["truncated"] == (ep["arrived"] < N)).all()
            # auto-reset inside the call: the arena is at step 0 of its next episode and nobody has arrived
            assert (env.vec.get(_lib.FLD_STEP_COUNT)[ep["arena"]] == 0).all()
            assert not info["agent_arrived"].reshape(A, N)[ep["arena"]].any()
        assert (info["collisions"] >= 0).all() and info["collisions"].shape == (A,)
    assert ends == 2 * A
    env5 = adapters.AgentVectorEnv(H.OracleVec(2, 3, "doorway", _params(3, 5)), new_step_api=True)
    o0, i0 = env5.reset()
    for s in range(5):
        obs, rew, term, trunc, info = env5.step(np.zeros((6,), np.float32))
    assert trunc.all() and not term.any()                    # the cap, not an arrival, ended it


def test_multi_agent_vector_env_protocols():
    A, N = 3, 4
    p = H.scenario_params("crowd", N, max_step=400)
    env = adapters.MultiAgentVectorEnv(H.OracleVec(A, N, "crowd", p, seed=2), per_agent_dones=True)
    obs = env.vector_reset()
    assert len(obs) == A and sorted(obs[0]) == ['agent_%d' % i for i in range(N)] and obs[0]['agent_0'].shape == (64,)
    seen_done = np.zeros((A, N), int)
    finished = np.zeros(A, bool)
    for s in range(400):
        acts = [{aid: [0.0] for aid in o} for o in obs]
        obs, rew, done, info = env.vector_step(acts)
        for e in range(A):
            assert set(rew[e]) == set(obs[e]) and set(done[e]) == set(obs[e]) | {'__all__'}
            assert set(info[e]) == set(obs[e]) | {'__common__'} and info[e]['__common__']['collisions'] >= 0
            for aid, d in done[e].items():
                if aid != '__all__' and d and not finished[e]:     # first episode of each sub-environment
                    seen_done[e, int(aid.split('_')[1])] += 1
            if done[e]['__all__'] and not finished[e]:
                finished[e] = True
                obs[e] = env.reset_at(e)                     # the caller resets a finished sub-environment
                assert len(obs[e]) == N
        if finished.all():
            break
    assert finished.all() and seen_done.max() == 1           # every agent reported done exactly once per episode
    with pytest.raises(KeyError):
        env.vector_step([{'agent_0': [0.0]}] * A)
    # BaseEnv protocol: poll -> send_actions -> poll
    env2 = adapters.MultiAgentVectorEnv(H.OracleVec(2, 3, "doorway", _params(3, 4)), per_agent_dones=False)
    o, r, d, i, off = env2.poll()
    assert sorted(o) == [0, 1] and off == {} and d[0]['__all__'] is False
    for s in range(4):
        env2.send_actions({e: {aid: [0.1] for aid in o[e]} for e in o})
        o, r, d, i, off = env2.poll()
    assert d[0]['__all__'] and d[0]['agent_0'] is False      # bug-compatible per-agent flags (env.py:470)
    assert i[0]['__common__']['truncated'] is True
    again = env2.try_reset(1)
    assert list(again) == [1] and len(again[1]) == 3
    assert env2.get_sub_environments() == []
