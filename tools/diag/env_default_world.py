#!/usr/bin/env python3
"""Diagnostic: throughput of the reference env's OWN configuration (env.py:26-123: doorway world, 10 agents, K = 5,
neighborDist 1.5) as a batch: A arenas, full step with observation, and ORCA-only rollout."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from collision_avoidance_amd import scenarios, _lib
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
for A in (256, 1024, 4096, 16384):
    env = VecCollisionAvoidanceEnv(A, 10, scenario="doorway", params=scenarios.env_params(), use_torch=True)
    pool = torch.rand((16, A, 10), device="cuda") - 0.5
    for mode in ("step", "orca"):
        def run(n):
            if mode == "step":
                for i in range(n):
                    env._call("ca_step", env.h, pool[i % 16].data_ptr(), _lib.F_STATS | _lib.F_OBS | _lib.F_AUTORESET)
            else:
                for i in range(n // 50):
                    env._call("ca_rollout", env.h, 50, _lib.F_STATS | _lib.F_AUTORESET)
        run(3000); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("doorway A=%5d %s: %.2f us/step, %.1f M agent-steps/s  %s" % (A, mode, dt / 2000 * 1e6, A * 10 * 2000 / dt / 1e6, env.launch_info()))
    env.close()
