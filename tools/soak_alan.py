#!/usr/bin/env python3
"""Diagnostic: long GPU-vs-oracle parity soak of the ALAN online step (ALAN_true.py:569-628) on batches large enough for every form
it takes -- inside the four-lanes kernel (one launch per 256 steps of a rollout), inside the one-lane kernels (obstacle lists of 4 and
16, many-obstacle agents solved apart), per-arena freezing at the episode's end (run_sim, ALAN:106-123) -- state, lists, fp64 weights
and times, actions, arrival steps and counters, bit for bit.  Usage (GPU box): python tools/soak_alan.py [steps] [seed offset]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from collision_avoidance_amd import _lib, alan
from oracle import oracle as o
from tests import helpers as H

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed_offset = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # other scenario draws (and, through them, other bandit draws)
ACTS9 = [(1, 0), (0.70711, 0.70711), (0, 1), (-0.70711, 0.70711), (-1, 0), (-0.70711, -0.70711), (0, -1), (0.70711, -0.70711), (0.5, 0.1)]
cases = [("crowd", 16, 512, alan.DEFAULT_ACTIONS, 1),      # four lanes per agent: the bandit inside a one-launch rollout
         ("crowd", 64, 2048, alan.DEFAULT_ACTIONS, 3),     # one lane per agent, one wave per arena
         ("circle", 100, 1024, ACTS9[:3], 5),              # two waves per arena (the reference's own ALAN run, ALAN:755-757)
         ("congested", 50, 2048, ACTS9, 7),                # obstacle lists of 16
         ("deadlock", 50, 2048, ACTS9[:2], 9),             # register lines by residency, many-obstacle agents eight at a time
         ("blocks", 20, 4096, alan.DEFAULT_ACTIONS, 11),   # a world per arena
         ("deadlock", 50, 1100, ACTS9, 13)]                # the LDS line table (a batch the chip holds at once)
total = 0
for scen, N, A, acts, seed in cases:
    seed += seed_offset
    t0 = time.time()
    p = H.scenario_params(scen, N, max_step=steps - 40)    # the cap ends the episodes of the slow arenas inside the run
    g = H.make_gpu(A, N, scen, p, seed=seed)
    e = H.make_oracle(A, N, scen, p, seed=seed)
    g.alan_configure(acts); e.alan_configure(acts)
    sc = (np.arange(A) % 29).astype(np.int32)              # arenas end at different steps
    g.set(_lib.FLD_STEP_COUNT, sc); e.set(o.FLD_STEP_COUNT, sc)
    done = 0
    while done < steps:
        n = min(100, steps - done)
        g.alan_rollout(n, stats=True, freeze=True)
        for s in range(n):
            e.alan_step(flags=o.F_STATS | o.F_FREEZE)
        done += n
        what = "%s N=%d step %d" % (scen, N, done)
        H.assert_state_equal(g, e, what, reward=True)
        for f, of in ((_lib.FLD_ALAN_WEIGHTS, o.FLD_ALAN_WEIGHTS), (_lib.FLD_ALAN_TIMES, o.FLD_ALAN_TIMES)):
            assert np.array_equal(g.get(f).view(np.uint64), e.get(of).view(np.uint64)), what + " weights / times"
        H._eq(g.get(_lib.FLD_ALAN_ACTION), e.get(o.FLD_ALAN_ACTION), what + " action")
        H._eq(g.get(_lib.FLD_ARRIVE_STEP), e.get(o.FLD_ARRIVE_STEP), what + " arrive_step")
    H.assert_stats_equal(g, e, scen)
    st = g.stats()
    total += st["agent_steps"]
    print("ok  %-9s A=%d N=%d actions=%d steps=%d  (%.1f s)  lanes/agent %d, lds %d  arenas done %d  stats %s" % (
        scen, A, N, len(acts), steps, time.time() - t0, g.launch_info()["lanes_per_agent"], g.launch_info()["lds_bytes"],
        int(g.get(_lib.FLD_ARENA_DONE).sum()), st), flush=True)
    g.close()
print("ALAN soak passed: %.3g agent-steps" % total)
