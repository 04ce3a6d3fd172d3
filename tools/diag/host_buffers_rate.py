#!/usr/bin/env python3
"""Diagnostic: the PCIe-inclusive rate of the full step at C3 when the caller hands over HOST buffers -- ca_step_host (actions
host -> device, 1 MB) and the observation, reward and done flags copied back to the host every step (67 MB + 1 MB) -- next to the
resident path that bench.py measures.  This figure is never bench.py's `value` (PCIe note)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from collision_avoidance_amd import scenarios
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
w = scenarios.BENCH_CONFIGS["C3"]
A, N = w["n_arenas"], w["n_agents"]
env = VecCollisionAvoidanceEnv(A, N, "crowd", scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]), use_torch=False)
rng = np.random.RandomState(0)
pool = [rng.uniform(-0.5, 0.5, (A, N)).astype(np.float32) for _ in range(4)]
env.rollout(2000, stats=True)     # settle the crowd
for mode in ("host actions in, observation / reward / done out (new pageable arrays)",
             "host actions in, observation / reward / done out (the environment's page-locked buffers, copy=False)",
             "host actions in, nothing copied back"):
    out = "done out" in mode
    pinned = "copy=False" in mode
    for i in range(5):
        env.step(pool[i % 4], with_obs=True, stats=True, copy=not pinned) if out else env._call("ca_step_host", env.h, pool[i % 4].ctypes.data, 3)
    env.sync()
    n = 100
    t0 = time.perf_counter()
    for i in range(n):
        if out:
            env.step(pool[i % 4], with_obs=True, stats=True, copy=not pinned)
        else:
            env._call("ca_step_host", env.h, pool[i % 4].ctypes.data, 3)   # CA_F_OBS | CA_F_STATS
    env.sync()
    dt = (time.perf_counter() - t0) / n
    print("C3 full step, %s: %.1f us per step = %.1f M agent-steps/s" % (mode, dt * 1e6, A * N / dt / 1e6))
