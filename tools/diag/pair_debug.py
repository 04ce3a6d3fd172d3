#!/usr/bin/env python3
"""Diagnostic: first step of a large arena, pair kernel against the oracle -- which agents differ, and what they have in common."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as o
from tests import helpers as H
from collision_avoidance_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = H.scenario_params("crowd", N)
gpu = H.make_gpu(2, N, "crowd", p, seed=3)
orc = H.make_oracle(2, N, "crowd", p, seed=3)
print(gpu.launch_info())
gpu.orca_step(stats=True); orc.orca_step(flags=o.F_STATS)
gc, gi = gpu.neighbor_lists(); oc, oi = orc.get(o.FLD_NB_COUNT), orc.get(o.FLD_NB_IDX)
print("nb_count equal:", np.array_equal(gc, oc))
K = oi.shape[2]
mask = np.arange(K)[None, None, :] < oc[:, :, None]
print("nb_idx equal:", np.array_equal(np.where(mask, gi, -1), np.where(mask, oi, -1)))
goc, goi = gpu.obstacle_neighbor_lists(); ooc = orc.get(o.FLD_OBST_COUNT)
print("obst_count equal:", np.array_equal(goc, ooc))
gv, ov = gpu.get(_lib.FLD_VEL_X), orc.get(o.FLD_VEL_X)
bad = gv != ov
print("vel_x differs at", bad.sum(), "of", bad.size)
print("ncnt histogram of bad:", np.bincount(oc[bad], minlength=11), " of all:", np.bincount(oc.reshape(-1), minlength=11))
print("obst count histogram of bad:", np.bincount(ooc[bad], minlength=5), " of all:", np.bincount(ooc.reshape(-1), minlength=5))
idx = np.argwhere(bad)[:8]
for a, i in idx:
    print(a, i, "ncnt", oc[a, i], "no", ooc[a, i], "gpu v", gv[a, i], gpu.get(_lib.FLD_VEL_Y)[a, i], "orc v", ov[a, i], orc.get(o.FLD_VEL_Y)[a, i])
