#!/usr/bin/env python3
"""Diagnostic: throughput of the C3 step over time from a cold start (DVFS: boost clocks for the first fraction of
a second, then the sustained clock).  Prints, for every window of 250 steps for about 4 s: agent-steps/s, the shader
clock the driver reports at the end of the window (sysfs pp_dpm_sclk, the level marked `*`; MHz) and the board power
(hwmon power1_average / power1_input; W) -- so that a falling rate can be told from a changing workload."""
import glob, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from collision_avoidance_amd import scenarios, _lib
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
w = scenarios.BENCH_CONFIGS["C3"]
A, N = w["n_arenas"], w["n_agents"]
env = VecCollisionAvoidanceEnv(A, N, scenario="crowd", params=scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]), use_torch=True)
pool = torch.rand((16, A, N), device="cuda") - 0.5


def _first(paths):
    for pat in paths:
        for f in sorted(glob.glob(pat)):
            try:
                return open(f).read()
            except Exception:
                pass
    return None


def sclk_mhz():
    t = _first(["/sys/class/drm/card*/device/pp_dpm_sclk"])
    if not t:
        return -1
    m = re.search(r"(\d+)\s*Mhz\s*\*", t, re.I)
    return int(m.group(1)) if m else -1


def power_w():
    t = _first(["/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"])
    try:
        return int(t) / 1e6
    except Exception:
        return -1.0


torch.cuda.synchronize()
print("idle: sclk %d MHz, %.0f W" % (sclk_mhz(), power_w()))
t_start = time.perf_counter()
out = []
for win in range(100):
    t0 = time.perf_counter()
    for i in range(250):
        env._call("ca_step", env.h, pool[i % 16].data_ptr(), _lib.F_STATS | _lib.F_OBS)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out.append((t1 - t_start, A * N * 250 / (t1 - t0) / 1e6, sclk_mhz(), power_w()))
print("time : M agent-steps/s : sclk MHz : W")
print(" ".join("%.2fs:%.0f:%d:%.0f" % o for o in out))
# the same state stepped again from the start at the sustained clock: is the late rate a property of the clock or of the
# crowd's state?  (the crowd of window 0 is replayed by a fresh environment while the chip is still warm)
env2 = VecCollisionAvoidanceEnv(A, N, scenario="crowd", params=scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]), use_torch=True)
torch.cuda.synchronize()
out2 = []
for win in range(12):
    t0 = time.perf_counter()
    for i in range(250):
        env2._call("ca_step", env2.h, pool[i % 16].data_ptr(), _lib.F_STATS | _lib.F_OBS)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out2.append((win, A * N * 250 / (t1 - t0) / 1e6, sclk_mhz(), power_w()))
print("fresh crowd on the warm chip (window : M agent-steps/s : sclk MHz : W)")
print(" ".join("%d:%.0f:%d:%.0f" % o for o in out2))
