#!/usr/bin/env python3
"""Diagnostic: where does a step-kernel (solve) wave spend its cycles?  Builds a CA_STAMPS variant of the
library into gpurun_out/ (never the product build), runs the bench workload for a few steps and
prints the mean share of each phase.  Usage (GPU box): python tools/stamps.py [C3|C2|C5] [step|orca]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from collision_avoidance_amd import build as b

out = os.path.join(ROOT, "gpurun_out", "libcaenv_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([b.hipcc()] + b.HIPCC_FLAGS + ["-DCA_STAMPS", "-o", out, b.SOURCES[0]])
b.LIB_PATH = out  # the loader reads this
from collision_avoidance_amd import _lib, scenarios
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
mode = sys.argv[2] if len(sys.argv) > 2 else "step"
w = scenarios.BENCH_CONFIGS[wl]
A, N = w["n_arenas"], w["n_agents"]
env = VecCollisionAvoidanceEnv(A, N, "crowd", scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]),
                               use_torch=False)
env.L.ca_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
rng = np.random.RandomState(0)
names = ["load+pref+stage", "list counts", "(unused)", "obst lines", "agent lines", "LP2", "LP3+integrate",
         "barrier+stats", "reward/pref", "done test+reduce", "tail sync+write"]
acc = []
for s in range(120):
    if mode == "step":
        env.step(rng.uniform(-0.5, 0.5, (A, N)).astype(np.float32), with_obs=False, stats=True)
    else:
        env.orca_step(stats=True)
    if s >= 100:
        nw = C.c_int32()
        buf = np.zeros((A * max(1, N // 64) * 2, 16), np.uint64)
        env._call("ca_debug_stamps", env.h, buf.ctypes.data, buf.shape[0], C.byref(nw))
        t = buf[:nw.value, :12].astype(np.int64)
        acc.append(np.diff(t, axis=1))
# the observation kernel of the last step
nw = C.c_int32()
obuf = np.zeros((A * ((N + 15) // 16) * 4, 16), np.uint64)
if mode == "step":
    env.observe()
    env._call("ca_debug_stamps", env.h, obuf.ctypes.data, -obuf.shape[0], C.byref(nw))
    to = np.diff(obuf[:nw.value, :9].astype(np.int64), axis=1)
    to = to[(obuf[:nw.value, 8] > 0)]
    onames = ["stage arena+lists", "barrier", "pre-pass (windows, pairs)", "barrier", "phase A (tasks)", "barrier",
              "phase B (winner)", "store"]
    print("obs_kernel: %d waves, mean cycles/wave %.0f" % (len(to), to.sum(axis=1).mean()))
    for k, n in enumerate(onames):
        print("  %-26s %8.0f cycles  %5.1f %%" % (n, to[:, k].mean(), 100 * to[:, k].mean() / to.sum(axis=1).mean()))
d = np.concatenate(acc)
tot = d.sum(axis=1)
print("%s %s: %d waves sampled, mean cycles/wave %.0f (p50 %.0f, p95 %.0f)" %
      (wl, mode, len(d), tot.mean(), np.median(tot), np.percentile(tot, 95)))
for k, n in enumerate(names):
    print("  %-18s %8.0f cycles  %5.1f %%   (p95 %6.0f)" % (n, d[:, k].mean(), 100 * d[:, k].mean() / tot.mean(),
                                                           np.percentile(d[:, k], 95)))
