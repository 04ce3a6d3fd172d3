#!/usr/bin/env python3
"""Diagnostic: the dispatch timeline of the C3 solve kernel -- where does the part of its duration go that no wave occupies?

Four timestamps per variant, all on the device-wide 100 MHz counter (s_memrealtime):
  fence0   a one-wave kernel queued right before the dispatch has ENDED  (the stream is ordered: the dispatch cannot begin earlier)
  first    the first wave of the dispatch executes its first instruction
  last     the last wave of the dispatch executes its last instruction
  fence1   a one-wave kernel queued right behind the dispatch BEGINS     (the dispatch, its cache write-back included, is over)
plus the dispatch's own duration from the start / stop timestamps of its packet (hipExtLaunchKernel events) and the spread of
wave starts and ends.  Variants: the solve kernel of the settled C3 crowd (built with stamps at wave start / end only:
-DCA_STAMPS=4, otherwise the product's code); an EMPTY kernel of the same grid, block and LDS size; the solve kernel with
its state stores non-temporal (-DCA_NT_STATE); the solve kernel with grid / 2 workgroups (half the batch).
Usage (GPU box): python tools/diag/timeline.py [warm steps]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from collision_avoidance_amd import build as b

WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
outdir = os.path.join(ROOT, "variants")   # git-ignored, but it travels to the GPU box: built where no GPU time is spent
os.makedirs(outdir, exist_ok=True)
variant = os.environ.get("CA_TL_VARIANT", "")


def build(tag, extra):
    out = os.path.join(outdir, "libcaenv_tl_%s.so" % tag)
    if not os.path.exists(out):
        subprocess.check_call([b.hipcc()] + b.HIPCC_FLAGS + ["-DCA_STAMPS=4"] + extra + ["-o", out, b.SOURCES[0]])
    return out


if not variant:     # parent: build the variants, run each in a child process (one library per process)
    libs = {"plain": build("plain", []), "nt": build("nt", ["-DCA_NT_STATE=1"])}
    for v in ("plain", "empty", "nt", "half"):
        env = dict(os.environ, CA_TL_VARIANT=v, CA_TL_LIB=libs["nt" if v == "nt" else "plain"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), str(WARM)], env=env)
    raise SystemExit(0)

b.LIB_PATH = os.environ["CA_TL_LIB"]
from collision_avoidance_amd import _lib, scenarios
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv

w = scenarios.BENCH_CONFIGS["C3"]
A, N = (w["n_arenas"] // 2 if variant == "half" else w["n_arenas"]), w["n_agents"]
env = VecCollisionAvoidanceEnv(A, N, "crowd", scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]), use_torch=False)
L = env.L
L.ca_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
L.ca_debug_clock.argtypes = [C.c_void_p, C.c_int32]
L.ca_debug_clock_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
L.ca_debug_empty.argtypes = [C.c_void_p]
rng = np.random.RandomState(0)
pool = rng.uniform(-0.5, 0.5, (16, A, N)).astype(np.float32)
import torch   # device-resident actions: the measured launches must not wait for a host copy
acts = torch.as_tensor(pool).cuda()
torch.cuda.synchronize()
for s in range(WARM):
    env._call("ca_step", env.h, C.c_void_p(acts[s % 16].data_ptr()), _lib.F_STATS)
env.sync()
env.profile(1)
rows = []
for rep in range(40):
    # a step in front keeps the GPU busy while the host queues the fence, the measured dispatch and the second fence
    env._call("ca_step", env.h, C.c_void_p(acts[rep % 16].data_ptr()), _lib.F_STATS)
    env._call("ca_debug_clock", env.h, 0)
    if variant == "empty":
        env._call("ca_debug_empty", env.h)
    else:
        env._call("ca_step", env.h, C.c_void_p(acts[(rep + 1) % 16].data_ptr()), _lib.F_STATS)
    env._call("ca_debug_clock", env.h, 1)
    env.sync()
    clk = np.zeros((2, 2), np.uint64)
    env._call("ca_debug_clock_read", env.h, clk.ctypes.data, 2)
    nw = C.c_int32()
    buf = np.zeros((A * 4, 16), np.uint64)
    env._call("ca_debug_stamps", env.h, buf.ctypes.data, buf.shape[0], C.byref(nw))
    st, en = buf[:nw.value, 12].astype(np.int64), buf[:nw.value, 11].astype(np.int64)
    f0, f1 = int(clk[0, 1]), int(clk[1, 0])
    ms = env.profile_read()["reset_kernels" if variant == "empty" else "step_kernel"]
    if rep >= 8:
        rows.append([st.min() - f0, en.max() - f0, f1 - f0, ms[1] * 1e5 / max(1, 1)] +
                    [np.percentile(st - f0, q) for q in (10, 50, 90)] + [np.percentile(en - f0, q) for q in (10, 50, 90)] +
                    [(en - st).mean()])
r = np.array(rows, np.float64) / 100.0    # 10 ns ticks -> us
m = np.median(r, axis=0)
print("%-6s grid %5d x %d lanes | fence0 -> first wave %5.2f us | first -> last wave end %6.2f us | last wave end -> fence1 %5.2f us | "
      "fence0 -> fence1 %6.2f us | dispatch (packet timestamps, mean of the step launches) %6.2f us" % (
          variant, env.launch_info()["grid"], env.launch_info()["block"], m[0], m[1] - m[0], m[2] - m[1], m[2], m[3]))
print("       wave starts p10/p50/p90 after fence0: %5.2f %5.2f %5.2f us | wave ends p10/p50/p90: %5.2f %5.2f %5.2f us | mean wave lifetime %5.2f us" % tuple(m[4:11]))
env.close()
