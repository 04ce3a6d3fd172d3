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
import bench as _bench   # host_cores(): affinity capped by the cgroup CPU quota
cores = _bench.host_cores()
rng = np.random.RandomState(0)


def run(env, A, N, mode, threads, warm=20):
    """Thread pool: every thread steps its own block of arenas through the whole sample, no per-step join
    (oracle/ca_oracle.cpp orc_env_rollout_mt); the sample is sized from a short probe to last about `budget` seconds."""
    acts = rng.uniform(-0.5, 0.5, (8, A, N)).astype(np.float32) if mode == "step" else None
    flags = o.F_OBS if mode == "step" else 0
    env.rollout_mt(warm, acts, flags=flags, n_threads=threads)
    t0 = time.perf_counter()
    env.rollout_mt(8, acts, flags=flags, n_threads=threads)
    per_step = max(1e-7, (time.perf_counter() - t0) / 8)
    steps = int(max(16, min(200000, budget / per_step)))
    t0 = time.perf_counter()
    env.rollout_mt(steps, acts, flags=flags, n_threads=threads)
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
        for th in sorted(set([min(16, cores), cores])):   # 16 = the CPU share of a one-GPU job on this pool
            Am = th * max(1, min(16, 1024 // N))   # every thread owns a block of whole arenas
            rows.append(dict(config="%s %dx%d" % (wl, w["n_arenas"], N), mode=mode,
                             **run(H.make_oracle(Am, N, "crowd", p, seed=0), Am, N, mode, th)))
cpu = ""
try:
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
out = dict(kind="CPU restatement baseline (oracle/ca_oracle.cpp -O2, fp32, -ffp-contract=off; not Python-RVO2)",
           host_cpu=cpu, cores_used=cores, seconds_per_cell=budget,
           note="thread pool without a per-step join (orc_env_rollout_mt); sampled arenas of the same workload (the sample sizes are in each row); full = obs on (ORC_F_OBS)", rows=rows)
json.dump(out, open(sys.argv[1], "w"), indent=1)
for r in rows:
    print("%-32s %-5s threads %-3d  %12.0f agent-steps/s  (%d arenas x %d steps)" % (
        r["config"], r["mode"], r["threads"], r["agent_steps_per_s"], r["arenas"], r["steps"]))
