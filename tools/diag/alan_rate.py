#!/usr/bin/env python3
"""Diagnostic: ALAN online rollouts (ca_alan_rollout with CA_F_STATS), agent-steps/s, for a few shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from collision_avoidance_amd import scenarios, alan
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
SHAPES = (("crowd", 1024, 16), ("crowd", 1024, 50), ("circle", 1024, 100), ("crowd", 4096, 64))
if len(sys.argv) > 1 and sys.argv[1] == "worlds":   # the reference's ALAN runs that have obstacles (ALAN:738-772), as batches of lane-kernel size
    SHAPES = (("congested", 4096, 50), ("deadlock", 4096, 50), ("blocks", 8192, 20))
for scen, A, N in SHAPES:
    p = scenarios.alan_params(N, scen)
    if scen in ("crowd", "circle"):
        p.update(done_mode=scenarios.DONE_REGOAL, max_step=0)
    else:
        p.update(max_step=0)   # (no episode end: an agent that reaches its goal goes on to its second target and stays there, ALAN:553-562)
    env = VecCollisionAvoidanceEnv(A, N, scenario=scen, params=p, use_torch=False)
    env.alan_configure(alan.DEFAULT_ACTIONS)
    env.alan_rollout(1000, stats=True, freeze=False); env.sync()
    t0 = time.perf_counter(); env.alan_rollout(2000, stats=True, freeze=False); env.sync(); dt = time.perf_counter() - t0
    print("alan %s A=%d N=%d fused=%s: %.2f us/step, %.1f M agent-steps/s  %s" % (scen, A, N, os.environ.get("CA_ALAN_FUSED", "1"), dt / 2000 * 1e6, A * N * 2000 / dt / 1e6, env.launch_info()))
    env.close()
