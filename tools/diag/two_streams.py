#!/usr/bin/env python3
"""Diagnostic: does the observation of one half-batch overlap the solve of the other?  The C3 batch as ONE handle (4096 arenas, one
stream: solve and observation strictly one after the other) against TWO handles of 2048 arenas on their own streams, stepped
alternately by one host thread with no synchronisation between steps (each half still does whole steps: actions -> solve -> observation).
Usage (GPU box): python tools/diag/two_streams.py [steps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from collision_avoidance_amd import _lib, scenarios
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
w = scenarios.BENCH_CONFIGS["C3"]
N = w["n_agents"]
p = scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"])


def run(parts, label):
    A = w["n_arenas"] // parts
    envs = [VecCollisionAvoidanceEnv(A, N, "crowd", p, seed=0, arena_offset=k * A, use_torch=False) for k in range(parts)]
    gen = torch.Generator(device="cuda").manual_seed(1)
    pools = [(torch.rand((16, A, N), device="cuda", generator=gen) - 0.5) for _ in envs]
    torch.cuda.synchronize()

    def step(i):
        for e, pl in zip(envs, pools):
            e._call("ca_step", e.h, C.c_void_p(pl[i % 16].data_ptr()), _lib.F_STATS | _lib.F_OBS)
    for i in range(8000):
        step(i)
    for e in envs:
        e.sync()
    t0 = time.perf_counter()
    for i in range(STEPS):
        step(8000 + i)
    for e in envs:
        e.sync()
    dt = time.perf_counter() - t0
    print("%-40s %7.2f us per step of the whole batch = %7.1f M agent-steps/s" % (label, dt / STEPS * 1e6, w["n_arenas"] * N * STEPS / dt / 1e6))
    for e in envs:
        e.close()


run(1, "one handle, 4096 arenas, one stream")
run(2, "two handles x 2048 arenas, two streams")
run(4, "four handles x 1024 arenas, four streams")
run(1, "one handle again")
