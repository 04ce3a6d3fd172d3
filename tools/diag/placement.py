#!/usr/bin/env python3
"""Diagnostic: per-SIMD load of the C3 solve kernel and what a balanced block order gains (round 5, settled crowd).

profiles/r05_a_step_kernel_timeline.txt: the kernel's tail is the slowest of 1024 SIMDs, each dealt four arenas.  Workgroups b,
b + 1024, b + 2048, b + 3072 share a SIMD (checked below from HW_ID / XCC_ID).  The tool
  (1) estimates every arena's work by least squares from the finish times of the SIMD groups under several random block orders
      (the yardstick: what a perfect estimate could gain),
  (2) installs orders that fold the arenas, sorted by an estimate, over the groups -- INSIDE each residue class mod 8, so an arena
      stays on the XCD whose L2 the observation kernel reads it from (ca_obs.h) --
  (3) compares the solve kernel's duration (packet timestamps) under each: identity, fitted work, and the estimates a kernel could
      produce on the device: LP3 lanes of the previous step, neighbour-count sum, overlapping pairs.
CA_STAMPS=3 build in variants/ (never the product): two stamps per wave + HW_ID + the LP3 lanes / rounds of the wave.
Usage (GPU box): python tools/diag/placement.py [warm steps]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from collision_avoidance_amd import build as b

WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
out = os.path.join(ROOT, "variants", "libcaenv_place.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.path.exists(out):
    subprocess.check_call([b.hipcc()] + b.HIPCC_FLAGS + ["-DCA_STAMPS=3", "-o", out, b.SOURCES[0]])
if len(sys.argv) > 2 and sys.argv[2] == "build":
    raise SystemExit(0)
b.LIB_PATH = out
from collision_avoidance_amd import _lib, scenarios
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
import torch

w = scenarios.BENCH_CONFIGS["C3"]
A, N = w["n_arenas"], w["n_agents"]
env = VecCollisionAvoidanceEnv(A, N, "crowd", scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]), use_torch=False)
env.L.ca_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
env.L.ca_debug_set_order.argtypes = [C.c_void_p, C.c_void_p]
env.L.ca_debug_set_order.restype = C.c_int
rng = np.random.RandomState(int(os.environ.get('CA_PLACE_SEED', '0')))
acts = torch.as_tensor(rng.uniform(-0.5, 0.5, (16, A, N)).astype(np.float32)).cuda()
torch.cuda.synchronize()
G = 1024
step_no = [0]


def steps(n, flags=_lib.F_STATS):
    for _ in range(n):
        env._call("ca_step", env.h, C.c_void_p(acts[step_no[0] % 16].data_ptr()), flags)
        step_no[0] += 1


def read():
    env.sync()
    nw = C.c_int32()
    buf = np.zeros((A, 16), np.uint64)
    env._call("ca_debug_stamps", env.h, buf.ctypes.data, buf.shape[0], C.byref(nw))
    start, end = buf[:, 12].astype(np.int64), buf[:, 11].astype(np.int64)
    t0 = start.min()
    fin = (end - t0).reshape(4, G).max(axis=0) / 100.0          # group g = blocks g, g+1024, g+2048, g+3072
    lp3 = buf[:, 4].astype(np.int64)
    return fin, (end.max() - t0) / 100.0, (lp3 & 0xFFFFFFFF) - 1, lp3 >> 32, buf


def kernel_us(n=64, full=False):
    """Mean duration of the solve kernel (and the observation kernel) over n steps, from the packets' own timestamps."""
    env.sync(); env.profile(1); env.profile_read()
    steps(n, _lib.F_STATS | (_lib.F_OBS if full else 0))
    r = env.profile_read()
    env.profile(0)
    return r["step_kernel"][1] * 1e3, r["obs_kernel"][1] * 1e3


def set_order(order):
    o = None if order is None else np.ascontiguousarray(order, np.int32)
    rc = env.L.ca_debug_set_order(env.h, None if o is None else o.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc


steps(WARM)
fin, span, r2, lanes, buf = read()
hw, xcc = buf[:, 2].astype(np.int64), buf[:, 3].astype(np.int64) & 0xF
simd = (xcc << 20) | (hw & 0xFF30)          # XCC | SE, SH, CU (bits 8-15) | SIMD (bits 4-5)
same = sum(len(set(simd[g::G].tolist())) == 1 for g in range(G))
print("placement: %d distinct SIMDs; %d of %d groups {b, b+1024, b+2048, b+3072} sit on ONE SIMD; XCC of block b == b %% 8 for %.1f %% of the blocks" % (
    len(set(simd.tolist())), same, G, 100.0 * np.mean(xcc == (np.arange(A) % 8))))
base = kernel_us()
print("identity order: solve kernel %.2f us (span first wave -> last wave end of the last launch %.1f us); group finish p10/p50/p90/max %s" % (
    base[0], span, [round(float(np.percentile(fin, q)), 1) for q in (10, 50, 90, 100)]))
print("LP3: waves with 0/1/2/3+ pool rounds %s; infeasible lanes per wave mean %.1f" % (
    [int((r2 == k).sum()) for k in (0, 1, 2)] + [int((r2 >= 3).sum())], lanes.mean()))


def class_perm(rng_):
    """A random order that keeps block % 8 == arena % 8."""
    order = np.empty(A, np.int32)
    for r in range(8):
        idx = np.arange(r, A, 8)
        order[idx] = rng_.permutation(idx)
    return order


# (1) least squares: finish time of group g under order o = sum of the works of its four arenas
import scipy.sparse as sp
import scipy.sparse.linalg as spl
orders = []
for t in range(12):
    order = class_perm(rng)
    set_order(order)
    steps(3)
    f, s_, rr, ln, _ = read()
    orders.append((order, f, rr, ln))
data, ri, ci, y = [], [], [], []
for k, (order, f, rr, ln) in enumerate(orders):
    for q in range(4):
        blocks = np.arange(G) + q * G
        ri.append(np.arange(G) + k * G); ci.append(order[blocks]); data.append(np.ones(G))
    y.append(f)
M = sp.csr_matrix((np.concatenate(data), (np.concatenate(ri), np.concatenate(ci))), shape=(len(orders) * G, A))
wgt = spl.lsqr(M, np.concatenate(y), damp=0.05)[0]
pred = M @ wgt
print("least squares over %d random orders: residual rms %.2f us of mean group finish %.1f us; arena work p10/p50/p90/max = %s us" % (
    len(orders), float(np.sqrt(np.mean((pred - np.concatenate(y)) ** 2))), float(np.mean(np.concatenate(y))),
    [round(float(np.percentile(wgt, q)), 1) for q in (10, 50, 90, 100)]))


def per_arena(order, v):          # a per-block quantity of the last launch -> per arena
    out = np.zeros(A)
    out[order] = v
    return out


order, f, rr, ln = orders[-1]
a_lanes, a_rounds = per_arena(order, ln), per_arena(order, rr)
cnt = env.get(_lib.FLD_NB_COUNT).sum(axis=1).astype(np.float64)
st0 = env.get(_lib.FLD_ARENA_STATS)[:, 1].astype(np.float64)
steps(16)
env.sync()
pairs = env.get(_lib.FLD_ARENA_STATS)[:, 1].astype(np.float64) - st0
for nm, v in (("LP3 lanes", a_lanes), ("LP3 rounds", a_rounds), ("neighbour-count sum", cnt), ("overlapping pairs / 16 steps", pairs)):
    print("correlation of the fitted arena work with %-28s %.3f" % (nm, float(np.corrcoef(wgt, v)[0, 1]) if v.std() > 0 else 0.0))
X = np.stack([np.ones(A), a_lanes, a_rounds, cnt, pairs], axis=1)
coef = np.linalg.lstsq(X, wgt, rcond=None)[0]
print("linear model of the fitted work on (1, LP3 lanes, LP3 rounds, count sum, pairs): coefficients %s, correlation %.3f" % (
    np.round(coef, 4).tolist(), float(np.corrcoef(X @ coef, wgt)[0, 1])))


def folded(cost):
    """Heaviest with lightest, inside each residue class mod 8 (512 arenas over 128 groups of four)."""
    order = np.empty(A, np.int32)
    for r in range(8):
        ar = np.arange(r, A, 8)                      # the arenas (and blocks) of this class
        s = ar[np.argsort(-cost[ar], kind="stable")]
        n = len(ar) // 4                             # groups of the class: blocks r + 8 j (+ 1024 q)
        for j in range(n):
            g = r + 8 * j
            order[g] = s[j]; order[g + G] = s[2 * n - 1 - j]; order[g + 2 * G] = s[2 * n + j]; order[g + 3 * G] = s[4 * n - 1 - j]
    return order


results = []
for name, cost in (("fitted work (yardstick)", wgt), ("LP3 lanes", a_lanes + 1e-3 * cnt), ("LP3 rounds, then lanes", a_rounds * 100 + a_lanes),
                   ("neighbour-count sum", cnt), ("overlapping pairs", pairs), ("the linear model", X @ coef), ("random (class-preserving)", rng.rand(A))):
    o_ = folded(cost)
    pair = []
    for rep in range(3):                 # the order and the identity alternately: the crowd drifts, the difference does not
        set_order(o_); steps(4); us = kernel_us(48)[0]
        f, s_, _, _, _ = read()
        set_order(None); steps(4); ident = kernel_us(48)[0]
        pair.append((us, ident))
    d = [100.0 * (u / i - 1.0) for u, i in pair]
    results.append((name, float(np.mean(d))))
    print("order folded by %-28s: solve kernel %s us against identity %s us: %+.1f %% (%s); group finish p10/p50/p90/max %s" % (
        name, [round(u, 2) for u, _ in pair], [round(i, 2) for _, i in pair], float(np.mean(d)), ", ".join("%+.1f" % x for x in d),
        [round(float(np.percentile(f, q)), 1) for q in (10, 50, 90, 100)]))
set_order(None)
again = kernel_us()
print("identity again: solve kernel %.2f us" % again[0])
# the full step under the best device-side order: does the observation kernel keep its XCD locality?
full_id = kernel_us(64, full=True)
set_order(folded(a_lanes + 1e-3 * cnt))
steps(4)
full_bal = kernel_us(64, full=True)
print("full step, identity: solve %.2f + observation %.2f us; folded by LP3 lanes (stale by now): solve %.2f + observation %.2f us" % (
    full_id + full_bal))
env.close()
