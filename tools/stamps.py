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

RT = len(sys.argv) > 3 and sys.argv[3] == "rt"   # wall-clock stamps (10 ns ticks) for wave timelines
# variants/ is git-ignored but travels to the GPU box: `python tools/stamps.py build` here spends no GPU time on the compiler
out = os.path.join(ROOT, "variants", "libcaenv_stamps%d.so" % (2 if RT else 1))
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.path.exists(out) or any(os.path.getmtime(f) > os.path.getmtime(out) for f in b.SOURCES):
    subprocess.check_call([b.hipcc()] + b.HIPCC_FLAGS + ["-DCA_STAMPS=%d" % (2 if RT else 1), "-o", out, b.SOURCES[0]])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    raise SystemExit(0)
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
if env.launch_info()["lanes_per_agent"] == 4:   # the quad kernel stamps its own phases (csrc/ca_quad.h)
    names = ["pref+stage arena", "obstacle nbrs+merge", "agent scan+merge+lists", "obst lines", "agent lines", "LP2",
             "LP3+integrate", "barriers+stats", "reward/pref", "done test+reduce", "reset/orient/tail"]
if env.launch_info()["lanes_per_agent"] == 2:   # the pair kernel stamps its own phases (csrc/ca_pair.h)
    names = ["load+pref+stage", "grid build+remap", "obstacle nbrs", "grid scan+merge", "lists+obst lines", "agent lines", "LP2",
             "LP3+integrate", "wait at the barrier", "2 barriers+stats", "reward+done+tail"]
acc, nacc = [], []
WARM = int(os.environ.get("CA_STAMPS_WARM", "3000"))   # the crowd settles over the first seconds: stamp the settled one
env.rollout(WARM, stats=True)
for s in range(120):
    if mode == "step":
        env.step(rng.uniform(-0.5, 0.5, (A, N)).astype(np.float32), with_obs=False, stats=True)
    else:
        env.orca_step(stats=True)
    if s >= 100:
        nw = C.c_int32()
        buf = np.zeros((A * max(1, N // 16) * 4, 16), np.uint64)
        env._call("ca_debug_stamps", env.h, buf.ctypes.data, buf.shape[0], C.byref(nw))
        t = buf[:nw.value, :12].astype(np.int64)
        acc.append(np.diff(t, axis=1))
        nacc.append(np.diff(buf[:nw.value, 12:16].astype(np.int64), axis=1))
        if s == 119:   # launch stagger: when do waves start / end relative to the first wave of the kernel?
            for nm, a0, a1 in (("nbr_kernel", 12, 15), ("step_kernel", 0, 11)):
                st, en = buf[:nw.value, a0].astype(np.int64), buf[:nw.value, a1].astype(np.int64)
                if not RT:
                    break      # the shader-clock counters of different CUs are not synchronised
                t0 = st.min()
                dur = en - st
                print("%s wave duration p10/p50/p90/p99/max = %s; mean end per XCD (block %% 8) = %s; slowest 8 waves: blocks %s" % (
                    nm, [int(np.percentile(dur, q)) for q in (10, 50, 90, 99, 100)],
                    [int((en - t0)[np.arange(nw.value) % 8 == x].mean()) for x in range(8)], np.argsort(-(en - t0))[:8].tolist()))
                late = np.argsort(-(en - t0))[:int(0.02 * nw.value)]
                print("   the 2%% waves that end last: start p50 %d, duration p50 %d; block index mod 32 histogram %s" % (
                    np.median((st - t0)[late]), np.median(dur[late]), np.bincount(late % 32, minlength=32).tolist()))
                print("%s wave timeline (10 ns ticks; span %d): start p10/p50/p90/max = %s; end p10/p50/p90/max = %s" % (
                    nm, en.max() - t0, [int(np.percentile(st - t0, q)) for q in (10, 50, 90, 100)],
                    [int(np.percentile(en - t0, q)) for q in (10, 50, 90, 100)]))
# the observation kernel of the last step
nw = C.c_int32()
obuf = np.zeros((A * ((N + 15) // 16) * 4, 16), np.uint64)
if mode == "step":
    env.observe()
    env._call("ca_debug_stamps", env.h, obuf.ctypes.data, -obuf.shape[0], C.byref(nw))
    if RT:
        ok = obuf[:nw.value, 8] > 0
        st, en = obuf[:nw.value, 0][ok].astype(np.int64), obuf[:nw.value, 8][ok].astype(np.int64)
        print("obs_kernel wave timeline (10 ns ticks; span %d): start p10/p50/p90/max = %s; end p10/p50/p90/max = %s" % (
            en.max() - st.min(), [int(np.percentile(st - st.min(), q)) for q in (10, 50, 90, 100)],
            [int(np.percentile(en - st.min(), q)) for q in (10, 50, 90, 100)]))
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
if env.launch_info()["lanes_per_agent"] != 1:
    raise SystemExit(0)
nd = np.concatenate(nacc)
print("nbr_kernel: %d waves sampled, mean cycles/wave (first to last stamp) %.0f" % (len(nd), nd.sum(axis=1).mean()))
for k, n in enumerate(["load+stage+obstacle edges", "agent scan", "list stores"]):
    print("  %-30s %8.0f cycles  %5.1f %%   (p95 %6.0f)" % (n, nd[:, k].mean(), 100 * nd[:, k].mean() / nd.sum(axis=1).mean(),
                                                           np.percentile(nd[:, k], 95)))
