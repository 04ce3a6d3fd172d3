"""The differential scenes of tests/orca_scenes.py on the HIP kernels: every scene is one arena of ONE batched environment
(its own obstacle table, its agents padded to eight with far-away bystanders), advanced by one ORCA step and compared with
the oracle bit for bit -- new velocities, positions, agent- and obstacle-neighbour lists.  tests/test_oracle_orca_definition.py
compares the oracle on the SAME scenes with an independent fp64 restatement, branch by branch (every branch of SURVEY
App. A.3 / A.4 / A.5 taken 100+ times); this file closes the chain for each solve kernel: the register-line lane kernel with
its solved-apart path (more than four edges in range), the LDS-line-table kernel and the four-lanes kernel."""
import os

import numpy as np
import pytest

from tests import helpers as H
from tests import orca_scenes as S
from oracle import oracle as o

pytestmark = pytest.mark.gpu
N_PAD = 8


def _batch(scenes):
    A = len(scenes)
    pos = np.zeros((A, N_PAD, 2), np.float32)
    vel = np.zeros((A, N_PAD, 2), np.float32)
    pref = np.zeros((A, N_PAD, 2), np.float32)
    for a, sc in enumerate(scenes):
        n = len(sc["pos"])
        assert n <= N_PAD
        pos[a, :n], vel[a, :n], pref[a, :n] = sc["pos"], sc["vel"], sc["pref"]
        for k in range(n, N_PAD):          # bystanders: beyond every range, 100 apart
            pos[a, k] = (1000.0 + 100.0 * k, 2000.0)
    worlds = [[np.asarray(q, np.float32) for q in sc["polys"]] for sc in scenes]
    return pos, vel, pref, worlds


def _params():
    return dict(time_step=S.DT, neighbor_dist=S.NEIGHBOR_DIST, max_neighbors=N_PAD - 1, time_horizon=S.TAU,
                time_horizon_obst=S.TAU_OBST, radius=S.R, max_speed=S.VMAX, max_step=0, done_mode=1, done_x_thresh=0.0,
                reward_scale=0.3, spawn_x0=0.0, spawn_x1=1.0, spawn_y0=0.0, spawn_y1=1.0, goal_x0=0.0, goal_x1=1.0,
                goal_y0=0.0, goal_y1=1.0)


@pytest.mark.parametrize("kernel", ["lane", "table", "quad"])
def test_differential_scenes_gpu_equals_oracle(kernel):
    from collision_avoidance_amd import _lib
    from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
    scenes = S.all_scenes(1.0)
    pos, vel, pref, worlds = _batch(scenes)
    A = len(scenes)
    p = _params()
    over = {"lane": {"CA_QUAD": "0"}, "table": {"CA_QUAD": "0", "CA_REG_LINES": "0"}, "quad": {"CA_QUAD": "1"}}[kernel]
    old = {k: os.environ.get(k) for k in over}
    os.environ.update(over)
    try:
        g = VecCollisionAvoidanceEnv(A, N_PAD, scenario=None, params=p, seed=0, max_obst_neighbors=16, use_torch=False,
                                     obstacles=dict(per_arena=worlds))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    info = g.launch_info()
    assert info["lanes_per_agent"] == (4 if kernel == "quad" else 1), info
    c = o.OracleEnv(o.make_config(n_arenas=A, n_agents=N_PAD, seed=0, max_obst_neighbors=16, **p))
    c.set_obstacles_per_arena(worlds)
    goal = (pos + pref).astype(np.float64)
    for env, F in ((g, _lib), (c, o)):
        env.set(F.FLD_POS_X, pos[..., 0]); env.set(F.FLD_POS_Y, pos[..., 1])
        env.set(F.FLD_VEL_X, vel[..., 0]); env.set(F.FLD_VEL_Y, vel[..., 1])
        env.set(F.FLD_PREF_X, pref[..., 0]); env.set(F.FLD_PREF_Y, pref[..., 1])
        env.set(F.FLD_GOAL_X, goal[..., 0]); env.set(F.FLD_GOAL_Y, goal[..., 1])
        env.set(F.FLD_GOAL2_X, goal[..., 0]); env.set(F.FLD_GOAL2_Y, goal[..., 1])
    g.orca_step(stats=True, no_done=True)
    c.orca_step(flags=o.F_STATS | o.F_NODONE)
    g.sync()                                     # (no world here has more than 16 edges in range: the overflow status stays clear)
    H.assert_state_equal(g, c, "scenes/" + kernel)
    H.assert_stats_equal(g, c, "scenes/" + kernel)
    # the step did something in every family: the focus agent's velocity changed from its preferred one somewhere
    nv = np.stack([g.get(_lib.FLD_VEL_X)[:, 0], g.get(_lib.FLD_VEL_Y)[:, 0]], 1)
    fam = np.array([sc["family"] for sc in scenes])
    for f in sorted(set(fam)):
        m = fam == f
        assert np.mean(np.linalg.norm(nv[m] - pref[m, 0], axis=1) > 1e-3) > 0.3, f
    g.close()
