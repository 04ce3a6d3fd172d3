#!/usr/bin/env python3
"""Diagnostic: throughput of the C3 step over time from a cold start (DVFS: boost clocks for the first fraction of
a second, then the sustained clock).  Prints agent-steps/s of every window of 250 steps for about 4 s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from collision_avoidance_amd import scenarios, _lib
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
w = scenarios.BENCH_CONFIGS["C3"]
A, N = w["n_arenas"], w["n_agents"]
env = VecCollisionAvoidanceEnv(A, N, scenario="crowd", params=scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]), use_torch=True)
pool = torch.rand((16, A, N), device="cuda") - 0.5
torch.cuda.synchronize()
t_start = time.perf_counter()
out = []
for win in range(100):
    t0 = time.perf_counter()
    for i in range(250):
        env._call("ca_step", env.h, pool[i % 16].data_ptr(), _lib.F_STATS | _lib.F_OBS)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out.append((t1 - t_start, A * N * 250 / (t1 - t0) / 1e6))
print(" ".join("%.2fs:%.0f" % o for o in out))
