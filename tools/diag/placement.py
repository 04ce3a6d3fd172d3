#!/usr/bin/env python3
"""Diagnostic: per-SIMD load of step_kernel at C3 and what a balanced block order would gain.
Workgroups b, b+1024, b+2048, b+3072 share a SIMD (measured, HW_ID); a SIMD is done when the work of its four
waves is done.  The tool (1) estimates every arena's work by least squares from the finish times of the SIMD
groups under several random block orders (an arena's work is persistent: agents move 1/60 per step), (2) installs
the order that folds the sorted arenas over the groups, (3) compares the kernel span.  CA_STAMPS=3 build in
gpurun_out/ (never the product)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from collision_avoidance_amd import build as b

out = os.path.join(ROOT, "gpurun_out", "libcaenv_place.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([b.hipcc()] + b.HIPCC_FLAGS + ["-DCA_STAMPS=3", "-o", out, b.SOURCES[0]])
b.LIB_PATH = out
from collision_avoidance_amd import _lib, scenarios
from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv

w = scenarios.BENCH_CONFIGS["C3"]
A, N = w["n_arenas"], w["n_agents"]
env = VecCollisionAvoidanceEnv(A, N, "crowd", scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"]), use_torch=False)
env.L.ca_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
env.L.ca_debug_set_order.argtypes = [C.c_void_p, C.c_void_p]
env.L.ca_debug_set_order.restype = C.c_int
rng = np.random.RandomState(0)
G = 1024


def step_and_read(n=1):
    for _ in range(n):
        env.step(rng.uniform(-0.5, 0.5, (A, N)).astype(np.float32), with_obs=False, stats=True)
    nw = C.c_int32()
    buf = np.zeros((A, 16), np.uint64)
    env._call("ca_debug_stamps", env.h, buf.ctypes.data, buf.shape[0], C.byref(nw))
    start, end = buf[:, 12].astype(np.int64), buf[:, 11].astype(np.int64)
    t0 = start.min()
    fin = (end - t0).reshape(4, G).max(axis=0) / 100.0          # group g = blocks g, g+1024, g+2048, g+3072
    lp3 = buf[:, 4].astype(np.int64)
    return fin, (end.max() - t0) / 100.0, (lp3 & 0xFFFFFFFF) - 1, lp3 >> 32


def set_order(order):
    o = None if order is None else np.ascontiguousarray(order, np.int32)
    rc = env.L.ca_debug_set_order(env.h, None if o is None else o.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc


step_and_read(300)
base = [step_and_read(1) for _ in range(8)]
print("identity order: kernel span %.1f us (8 steps: %s); group finish p10/p50/p90/max %s" % (
    np.mean([x[1] for x in base]), [round(x[1], 1) for x in base],
    [round(float(np.percentile(base[-1][0], q)), 1) for q in (10, 50, 90, 100)]))
rounds2 = base[-1][2]; lanes = base[-1][3]
print("LP3: waves with 1/2/3+ pool rounds %s; lanes per wave mean %.1f" % ([int((rounds2 == k).sum()) for k in (1, 2)] + [int((rounds2 >= 3).sum())], lanes.mean()))
# (1) least squares: finish time of group g under order o = sum of the works of its four arenas
rows, rhs = [], []
orders = []
for t in range(10):
    order = rng.permutation(A).astype(np.int32)
    set_order(order)
    step_and_read(2)
    fin, span, r2, ln = step_and_read(1)
    orders.append((order, fin, r2, ln))
import scipy.sparse as sp
import scipy.sparse.linalg as spl
data, ri, ci, y = [], [], [], []
for k, (order, fin, r2, ln) in enumerate(orders):
    for q in range(4):
        blocks = np.arange(G) + q * G
        ri.append(np.arange(G) + k * G); ci.append(order[blocks]); data.append(np.ones(G))
    y.append(fin)
M = sp.csr_matrix((np.concatenate(data), (np.concatenate(ri), np.concatenate(ci))), shape=(len(orders) * G, A))
wgt = spl.lsqr(M, np.concatenate(y), damp=0.05)[0]
pred = M @ wgt
print("least squares over %d random orders: residual rms %.2f us of mean %.1f us; arena work p10/p50/p90/max = %s us" % (
    len(orders), float(np.sqrt(np.mean((pred - np.concatenate(y)) ** 2))), float(np.mean(np.concatenate(y))),
    [round(float(np.percentile(wgt, q)), 1) for q in (10, 50, 90, 100)]))
# is what the fit leaves over a property of the SIMD (persistent over the orders) or noise?
R = (np.concatenate(y) - pred).reshape(len(orders), G)
cc = np.corrcoef(R)
off = cc[~np.eye(len(orders), dtype=bool)]
gmean = R.mean(axis=0)
print("residual of the fit per SIMD group: correlation between two orders %.3f on average; persistent part (mean over orders) "
      "rms %.2f us, p1/p99 %s us" % (float(off.mean()), float(np.sqrt(np.mean(gmean ** 2))),
                                      [round(float(np.percentile(gmean, q)), 1) for q in (1, 99)]))
nw_ = C.c_int32(); hb = np.zeros((A, 16), np.uint64)
env._call("ca_debug_stamps", env.h, hb.ctypes.data, hb.shape[0], C.byref(nw_))
xcc = (hb[:G, 3].astype(np.int64) & 0xF)
print("   mean residual per XCC: %s us" % [round(float(gmean[xcc == x].mean()), 2) for x in range(8)])
# proxies: LP3 lanes / rounds of the arena (arena of block b under the last order)
order, fin, r2, ln = orders[-1]
arena_lanes = np.zeros(A); arena_lanes[order] = ln
arena_r2 = np.zeros(A); arena_r2[order] = r2
print("correlation of the fitted arena work with its LP3 lanes %.3f, with its LP3 rounds %.3f" % (
    float(np.corrcoef(wgt, arena_lanes)[0, 1]), float(np.corrcoef(wgt, arena_r2)[0, 1])))


def folded(cost):
    s = np.argsort(-cost)                      # heaviest first
    order = np.empty(A, np.int32)
    order[0:G] = s[0:G]                        # group g: rank g ...
    order[G:2 * G] = s[2 * G - 1:G - 1:-1]     # ... 2G-1-g ...
    order[2 * G:3 * G] = s[2 * G:3 * G]        # ... 2G+g ...
    order[3 * G:4 * G] = s[4 * G - 1:3 * G - 1:-1]   # ... 4G-1-g
    return order


def folded_speed(cost, offset):
    """greedy: heaviest arena first, always into the group that would finish earliest (its offset + work so far),
    four arenas per group"""
    import heapq
    s = np.argsort(-cost)
    heap = [(float(offset[g]), g) for g in range(G)]
    heapq.heapify(heap)
    members = [[] for _ in range(G)]
    for a_ in s:
        while True:
            t, g = heapq.heappop(heap)
            if len(members[g]) < 4:
                break
        members[g].append(int(a_))
        if len(members[g]) < 4:
            heapq.heappush(heap, (t + float(cost[a_]), g))
    order = np.empty(A, np.int32)
    for g in range(G):
        for q_, a_ in enumerate(members[g]):
            order[q_ * G + g] = a_
    return order


set_order(folded_speed(wgt, gmean))
step_and_read(2)
res = [step_and_read(1) for _ in range(8)]
print("greedy order on fitted work + the SIMDs' persistent offsets: kernel span %.1f us (%s); group finish p10/p50/p90/max %s" % (
    np.mean([x[1] for x in res]), [round(x[1], 1) for x in res],
    [round(float(np.percentile(res[-1][0], q)), 1) for q in (10, 50, 90, 100)]))
for name, cost in (("fitted work", wgt), ("LP3 lanes", arena_lanes + 0.01 * wgt), ("LP3 rounds then lanes", arena_r2 * 100 + arena_lanes)):
    set_order(folded(cost))
    step_and_read(2)
    res = [step_and_read(1) for _ in range(8)]
    print("order folded by %-22s: kernel span %.1f us (%s); group finish p10/p50/p90/max %s" % (
        name, np.mean([x[1] for x in res]), [round(x[1], 1) for x in res],
        [round(float(np.percentile(res[-1][0], q)), 1) for q in (10, 50, 90, 100)]))
set_order(None)
res = [step_and_read(1) for _ in range(4)]
print("identity again: kernel span %.1f us" % np.mean([x[1] for x in res]))
