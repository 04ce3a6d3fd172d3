#!/usr/bin/env python3
"""Diagnostic: throughput of the reference's ALAN worlds that have obstacles (ALAN:195-208 congested, 359-372 blocks, 418-455 deadlock)
as batches: full step with observation, and ORCA-only rollout.  CA_REG_LINES=0 / 1 forces the LDS line table / the register lines.
usage: world_rate.py [scenario:N:A ...]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from collision_avoidance_amd import build as _b
if os.environ.get("CA_LIB"):   # A/B against another build of the library (variants/)
    _b.LIB_PATH = os.path.abspath(os.environ["CA_LIB"])
from collision_avoidance_amd import scenarios, _lib
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
shapes = [s.split(":") for s in sys.argv[1:]] or [("deadlock", 50, 1024), ("deadlock", 50, 4096), ("blocks", 20, 2048), ("blocks", 20, 8192),
                                                   ("congested", 50, 4096)]
for scen, N, A in shapes:
    N, A = int(N), int(A)
    p = scenarios.alan_params(N, scen)
    p.update(max_step=600)   # episodes of 10 s, then auto-reset (the agents start again from new places)
    env = VecCollisionAvoidanceEnv(A, N, scenario=scen, params=p, use_torch=True)
    pool = torch.rand((16, A, N), device="cuda") - 0.5
    for mode in ("step", "orca"):
        def run(n):
            if mode == "step":
                for i in range(n):
                    env._call("ca_step", env.h, pool[i % 16].data_ptr(), _lib.F_STATS | _lib.F_OBS | _lib.F_AUTORESET)
            else:
                for i in range(n // 50):
                    env._call("ca_rollout", env.h, 50, _lib.F_STATS | _lib.F_AUTORESET)
        run(1000); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(1000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%s N=%d A=%5d %s lib=%s reg_lines=%s: %.2f us/step, %.1f M agent-steps/s  lds %d overflow %d" % (
            scen, N, A, mode, os.path.basename(os.environ.get("CA_LIB", "product")), os.environ.get("CA_REG_LINES", "-"), dt / 1000 * 1e6, A * N * 1000 / dt / 1e6,
            env.launch_info()["lds_bytes"], env.stats()["obst_overflow"]))
    env.close()
