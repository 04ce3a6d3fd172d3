#!/usr/bin/env python3
"""Diagnostic: where does the hardware dispatcher put the waves of a small launch?  CA_STAMPS=3 build (slots 2, 3 of a
wave's stamp record = HW_ID, XCC_ID); prints the histogram of waves per SIMD and per CU for the solve kernel of a
workload.  usage (GPU box): python tools/diag/quad_placement.py [C2] [lanes: 1|4]"""
import ctypes as C
import os
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from collision_avoidance_amd import build as b

out = os.path.join(ROOT, "gpurun_out", "libcaenv_stamps3.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([b.hipcc()] + b.HIPCC_FLAGS + ["-DCA_STAMPS=3", "-o", out, b.SOURCES[0]])
b.LIB_PATH = out
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
os.environ["CA_QUAD"] = "1" if (len(sys.argv) <= 2 or sys.argv[2] == "4") else "0"
from collision_avoidance_amd import scenarios
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
w = scenarios.BENCH_CONFIGS[wl]
A, N = w["n_arenas"], w["n_agents"]
env = VecCollisionAvoidanceEnv(A, N, "crowd", scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]), use_torch=False)
env.L.ca_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
print(env.launch_info())
for rep in range(3):
    env.rollout(20, stats=True)
    nw = C.c_int32()
    buf = np.zeros((A * max(1, N // 16) * 4, 16), np.uint64)
    env._call("ca_debug_stamps", env.h, buf.ctypes.data, buf.shape[0], C.byref(nw))
    hw, xcc = buf[:nw.value, 2].astype(np.int64), buf[:nw.value, 3].astype(np.int64) & 0xF
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    per_simd = Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
    per_cu = Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    print("launch %d: %d waves on %d SIMDs (waves per SIMD histogram %s), %d CUs (waves per CU histogram %s), per XCC %s" % (
        rep, nw.value, len(per_simd), dict(sorted(Counter(per_simd.values()).items())), len(per_cu),
        dict(sorted(Counter(per_cu.values()).items())), dict(sorted(Counter(xcc.tolist()).items()))))
