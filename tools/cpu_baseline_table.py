#!/usr/bin/env python3
"""the CPU restatement (oracle/ca_oracle.cpp -- NOT Python-RVO2) timed on this host for every
BASELINE configuration x {ORCA-only, full step} x {1 core, all cores}.  Test/bench infrastructure: it times the
oracle, never the product.   usage: python tools/cpu_baseline_table.py out.json [seconds_per_cell]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from collision_avoidance_amd import scenarios
from oracle import oracle as o
from tests import helpers as H

budget = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
try:
    cores = max(1, len(os.sched_getaffinity(0)))
except Exception:
    cores = os.cpu_count() or 1
cores = min(cores, 64)
rng = np.random.RandomState(0)


def run(env, A, N, mode, threads, warm=20):
    acts = rng.uniform(-0.5, 0.5, (8, A, N)).astype(np.float32)
    flags = o.F_OBS if mode == "step" else 0
    for s in range(warm):
        env.step_mt(acts[s % 8] if mode == "step" else None, flags=flags, n_threads=threads)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget:
        for s in range(5):
            env.step_mt(acts[(steps + s) % 8] if mode == "step" else None, flags=flags, n_threads=threads)
        steps += 5
    dt = time.perf_counter() - t0
    return dict(agent_steps_per_s=A * N * steps / dt, arenas=A, steps=steps, seconds=round(dt, 2), threads=threads)


rows = []
# C1: the reference's own CPU-runnable cases, one arena: circle swap 1 x 8 (ALAN parameters) and the env's doorway 1 x 10
for name, scen, N, p in (("C1 circle 1x8", "circle", 8, scenarios.alan_params(8, "circle")),
                         ("C1 doorway 1x10 (env.py main)", "doorway", 10, scenarios.env_params())):
    pp = dict(p); pp.update(max_step=0)
    for mode in ("orca", "step"):
        env = H.make_oracle(1, N, scen, pp, seed=0)
        r = run(env, 1, N, mode, 1, warm=100)
        rows.append(dict(config=name, mode=mode, **r))
for wl in ("C2", "C3", "C5"):
    w = scenarios.BENCH_CONFIGS[wl]
    N = w["n_agents"]
    p = scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"])
    for mode in ("orca", "step"):
        A1 = max(1, min(64, 4096 // N))
        rows.append(dict(config="%s %dx%d" % (wl, w["n_arenas"], N), mode=mode,
                         **run(H.make_oracle(A1, N, "crowd", p, seed=0), A1, N, mode, 1)))
        Am = max(cores, min(16 * cores, (4096 // N) * cores // 4 or cores))
        rows.append(dict(config="%s %dx%d" % (wl, w["n_arenas"], N), mode=mode,
                         **run(H.make_oracle(Am, N, "crowd", p, seed=0), Am, N, mode, cores)))
cpu = ""
try:
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
out = dict(kind="CPU restatement baseline (oracle/ca_oracle.cpp -O2, fp32, -ffp-contract=off; not Python-RVO2)",
           host_cpu=cpu, cores_used=cores, seconds_per_cell=budget,
           note="sampled arenas of the same workload (the sample sizes are in each row); full = obs on (ORC_F_OBS)", rows=rows)
json.dump(out, open(sys.argv[1], "w"), indent=1)
for r in rows:
    print("%-32s %-5s threads %-3d  %12.0f agent-steps/s  (%d arenas x %d steps)" % (
        r["config"], r["mode"], r["threads"], r["agent_steps_per_s"], r["arenas"], r["steps"]))
