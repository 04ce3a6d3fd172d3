#!/usr/bin/env python3
"""Diagnostic: long GPU-vs-oracle parity soak (bit-exact state, lists, observation, reward, counters) on larger
batches than the unit tests use.  Usage (GPU box): python tools/soak_parity.py [arenas] [steps] [seed offset] [case numbers]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from collision_avoidance_amd import scenarios
from oracle import oracle as o
from tests import helpers as H

A = int(sys.argv[1]) if len(sys.argv) > 1 else 192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seed_offset = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # other scenario draws and actions: an independent run
only = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else None   # case numbers (0-based): a subset
threads = max(1, min(32, len(os.sched_getaffinity(0))))
cases = [("crowd", 64, scenarios.bench_params(64, 5.0, 10), 11), ("crowd", 64, scenarios.bench_params(64, 5.0, 10), 12),
         ("circle", 64, H.scenario_params("circle", 64), 3), ("doorway", 10, H.scenario_params("doorway", 10), 5),
         ("crowd", 16, scenarios.bench_params(16, 1.5, 5), 7), ("deadlock", 30, H.scenario_params("deadlock", 30), 9),
         ("crowd", 256, scenarios.bench_params(256, 5.0, 10), 13),
         ("blocks", 12, H.scenario_params("blocks", 12), 15),            # a world per arena
         ("crowd_separated", 64, scenarios.bench_params(64, 5.0, 10), 17),
         # round 4: the two-lanes kernel (arenas of 129 .. 512 agents: grid order, scan bounded by the previous list), the
         # register-line kernel with obstacle lists of 16 (K = 10 world of 14 edges), a mid-size arena on the lane kernel
         ("crowd", 512, scenarios.bench_params(512, 5.0, 10), 19), ("crowd", 180, scenarios.bench_params(180, 5.0, 10), 21),
         ("congested", 24, H.scenario_params("congested", 24), 23), ("crowd", 100, scenarios.bench_params(100, 5.0, 10), 25),
         # round 4, second session: many-edge worlds in batches that are NOT resident with the LDS line table take the register lines,
         # their many-obstacle agents solved apart eight at a time (from 1281 workgroups: the tube at 50 agents per arena, blocks at
         # three 20-agent arenas per wave -- twice the arenas for that one)
         ("deadlock", 50, H.scenario_params("deadlock", 50), 27), ("blocks", 20, H.scenario_params("blocks", 20), 29, 2)]
for ci, (scen, N, p, seed, *mult) in enumerate(cases):
    if only is not None and ci not in only:
        continue
    seed += seed_offset
    t0 = time.time()
    A_case = (A if N <= 64 else max(8, A // 16 if N <= 256 else A // 64)) * (mult[0] if mult else 1)
    g = H.make_gpu(A_case, N, scen, p, seed=seed)
    e = H.make_oracle(A_case, N, scen, p, seed=seed)
    rng = np.random.RandomState(seed)
    g.reset(); e.reset()
    for s in range(steps):
        act = rng.uniform(-0.6, 0.6, (A_case, N)).astype(np.float32)
        g.step(act, with_obs=True, stats=True, autoreset=(s % 3 == 0))
        e.step_mt(act, flags=o.F_OBS | o.F_STATS | (o.F_AUTORESET if s % 3 == 0 else 0), n_threads=threads)
        if s % 50 == 49 or s == steps - 1:
            H.assert_state_equal(g, e, "%s N=%d step %d" % (scen, N, s), obs=True, reward=True)
    H.assert_stats_equal(g, e, scen)
    assert g.stats()["obst_overflow"] == 0, "an obstacle-neighbour list overflowed in %s" % scen
    print("ok  %-9s A=%d N=%d steps=%d seed=%d  (%.1f s)  stats %s" % (scen, A_case, N, steps, seed, time.time() - t0, g.stats()), flush=True)
    g.close()
print("soak passed")
