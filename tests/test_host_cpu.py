"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/ca_env.h declares, struct layouts match, errors are loud, scenario geometry restates the
reference's constants.  No compute call is made (there is no GPU here and no CPU fallback)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from tools import alan_actions

from collision_avoidance_amd import _lib, scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "ca_env.h")).read()


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(ca_[a-z_0-9]+)\s*\(", _header(), re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    for name in declared:
        assert hasattr(L, name), name


def test_enums_and_struct_layout_match_header():
    h = _header()
    fields = re.search(r"enum ca_field \{(.*?)CA_FLD__COUNT", h, re.S).group(1)
    names = re.findall(r"(CA_FLD_[A-Z0-9_]+)", fields)
    for i, n in enumerate(names):
        assert getattr(_lib, n[3:]) == i, n
    for macro in ("CA_OBS_DIM", "CA_MAX_NEIGHBORS", "CA_MAX_OBST_NEIGHBORS", "CA_MAX_AGENTS"):
        assert int(re.search(r"#define %s (\d+)" % macro, h).group(1)) == getattr(_lib, macro[3:])
    for macro, val in (("CA_F_OBS", _lib.F_OBS), ("CA_F_STATS", _lib.F_STATS), ("CA_F_AUTORESET", _lib.F_AUTORESET),
                       ("CA_F_NODONE", _lib.F_NODONE)):
        assert int(re.search(r"#define %s (\d+)u" % macro, h).group(1)) == val
    cfg = re.search(r"typedef struct ca_config \{(.*?)\} ca_config;", h, re.S).group(1)
    cfg = re.sub(r"/\*.*?\*/", "", cfg, flags=re.S)
    decl = []
    for ty, rest in re.findall(r"(int32_t|int64_t|uint64_t|double|float)\s+([^;]+);", cfg):
        decl += [(v.strip(), ty) for v in rest.split(",")]
    ctype = {"int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "double": C.c_double, "float": C.c_float}
    assert [(n, ctype[t]) for n, t in decl] == list(_lib.Config._fields_)
    assert C.sizeof(_lib.Config) == 112
    # the oracle's config mirrors it field for field (tests fill both from one dict)
    from oracle import oracle as o
    assert [f[0] for f in o.Config._fields_] == [f[0] for f in _lib.Config._fields_]


def test_no_cpu_fallback_is_loud():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
    with pytest.raises(RuntimeError, match="no HIP device"):
        VecCollisionAvoidanceEnv(2, 4, use_torch=False)
    from collision_avoidance_amd.envs import Collision_Avoidance_Env
    with pytest.raises(RuntimeError):
        Collision_Avoidance_Env(numAgents=3)


def test_argument_validation_without_gpu():
    L = _lib.load()
    h = C.c_void_p()
    bad = _lib.Config(n_arenas=0, n_agents=4, max_obst_neighbors=1, **scenarios.env_params())
    assert L.ca_create(C.byref(bad), 0, None, C.byref(h)) == -5 and not h.value
    assert b"out of range" in L.ca_last_error(None)
    bad = _lib.Config(n_arenas=1, n_agents=4, max_obst_neighbors=1, **dict(scenarios.env_params(), max_neighbors=17))
    assert L.ca_create(C.byref(bad), 0, None, C.byref(h)) == -5
    assert L.ca_create(None, 0, None, C.byref(h)) == -1
    # units and magnitudes (include/ca_env.h): the in-range division / square root of the kernels are exact for worlds of O(1)
    # units only, so a configuration outside the documented ranges is refused at the boundary, not silently inexact
    for field, val in (("radius", 5e-7), ("radius", 2e3), ("time_step", 1e-6), ("time_step", 60.0), ("max_speed", 0.0),
                       ("neighbor_dist", float("nan")), ("time_horizon", 1e4), ("spawn_x1", 3e5), ("goal_y0", float("inf"))):
        bad = _lib.Config(n_arenas=1, n_agents=4, max_obst_neighbors=1, **dict(scenarios.env_params(), **{field: val}))
        assert L.ca_create(C.byref(bad), 0, None, C.byref(h)) == -5 and not h.value, (field, val)
        assert field.encode() in L.ca_last_error(None) or b"box coordinate" in L.ca_last_error(None), L.ca_last_error(None)
    for macro in ("CA_MIN_LENGTH", "CA_MAX_LENGTH", "CA_MIN_TIME_STEP", "CA_MAX_TIME_STEP", "CA_MAX_COORD", "CA_MIN_EDGE"):
        assert re.search(r"#define %s " % macro, _header()), macro


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under collision_avoidance_amd/ may import it."""
    pkg = os.path.join(ROOT, "collision_avoidance_amd")
    banned = re.compile(r"(^\s*(from|import)\s+oracle\b|from\s+\.+\s*import\s+oracle|#include\s*[\"<].*oracle|"
                        r"libca_oracle|\borc_[a-z_]+\s*\(|rvo2_shim)", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not banned.search(src), os.path.join(dirpath, f)


def test_scenario_geometry():
    assert scenarios.obstacles("doorway", 10) == [
        [(-15.0, 0.0), (-15.0, 10), (10, 10), (10, 0.0)],
        [(2.0, 0.0), (2.5, 0.0), (2.5, 4.4), (2.0, 4.4)],
        [(2.0, 5.6), (2.5, 5.6), (2.5, 10.0), (2.0, 10.0)]]
    assert scenarios.crowd_envsize(64) == 16.0 and scenarios.crowd_envsize(16) == 8.0
    p = scenarios.bench_params(64, 5.0, 10)
    assert p["done_mode"] == scenarios.DONE_REGOAL and p["max_step"] == 0 and p["goal_x1"] == 16.0
    assert scenarios.alan_params(8, "circle")["max_step"] == int((10 / (1 / 60.)) * 8)
    with pytest.raises(ValueError):
        scenarios.obstacles("nope", 4)


def test_action_set_files(tmp_path):
    """.act files are the repr of a list of tuples (Train_ALAN_action_space.py:150-153)."""
    from collision_avoidance_amd import alan
    f = tmp_path / "crowd_actions.act"
    f.write_text("[(1, 0), (0.06130798855686512, -0.9981188959934139), (-0.6767380373137093, 0.7362238985884584)]")
    acts = alan_actions.load_actions(str(f))
    assert acts == [(1.0, 0.0), (0.06130798855686512, -0.9981188959934139), (-0.6767380373137093, 0.7362238985884584)]
    g = tmp_path / "out.act"
    alan_actions.save_actions(str(g), acts)
    assert alan_actions.load_actions(str(g)) == acts
    (tmp_path / "bad.act").write_text("[]")
    with pytest.raises(ValueError):
        alan_actions.load_actions(str(tmp_path / "bad.act"))


def test_no_hot_kernel_uses_scratch_memory():
    """A register spill in a solve / observation kernel is HBM traffic (round 1: 19 MB per launch) and a dependent
    memory round trip inside the LP: the compiler's own metadata must show 0 bytes of scratch for every hot kernel
    (the LDS-line-table variant keeps LP3's projected lines in a private array by design).  Compiles the device
    assembly once per source change (tools/kernel_resources.py, about a minute)."""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    kr.ensure_asm()
    rows = kr.parse()
    names = [r["name"] for r in rows]
    for want in ("step_kernel<10, 64, 4, true, 1, 4, false>", "step_kernel<10, 64, 4, true, 1, 4, true>", "quad_kernel<5, 64, 4, false>",
                 "quad_kernel<10, 256, 16, false>", "quad_kernel<5, 64, 4, true>", "pair_kernel<10, 512>", "step_kernel<5, 64, 4, true, 1, 16, false>"):
        assert any(n.startswith(want) for n in names), want
    bad = kr.spilling(rows)
    assert not bad, [(r["name"], r["scratch"], r["vgpr_spill"]) for r in bad]
    # EVERY instantiation pick_variant (ca_env.hip) can select for K <= 10 -- lane kernels with register lines and obstacle
    # lists of 4 or 16 at every workgroup size, with and without the ALAN bandit; the four-lanes kernel in all its shapes; the
    # two-lanes kernel -- is present and has no scratch at all (round 4 exempted the multi-wave K = 10 / S = 16 shapes:
    # 32 B, 40 spilled registers; the reference's "deadlock" world with more than 64 agents selects them, ALAN:418-455)
    want = ["step_kernel<%d, %d, 4, true, 1, %d, false>" % (k, bs, sm) for k in (5, 10) for bs in (64, 128, 256, 512, 1024) for sm in (4, 16)]
    want += ["step_kernel<%d, %d, 4, true, 1, %d, true>" % (k, bs, sm) for k in (5, 10) for bs in (64, 128) for sm in (4, 16)]
    want += ["step_kernel<%d, %d, 4, true, 2, 4, false>" % (k, bs) for k in (5, 10) for bs in (256, 512)]
    want += ["quad_kernel<%d, %d, %d, %s>" % (k, bs, sq, al) for k in (5, 10) for bs in (64, 128, 256, 512) for sq in (4, 16) for al in ("false", "true")]
    want += ["pair_kernel<10, 256>", "pair_kernel<10, 512>", "obs_kernel<256"]
    for w_ in want:
        k = [r for r in rows if r["name"].startswith(w_)]
        assert k, w_
        assert all(r["scratch"] == 0 and r["vgpr_spill"] == 0 for r in k), (w_, [(r["scratch"], r["vgpr_spill"]) for r in k])
    # the only kernels with a scratch segment are the LDS-line-table ones (K > 10, or a many-edge world in a batch small enough
    # to be resident with the table: LP3's projected lines are a private array there by design)
    assert all(re.match(r"step_kernel<\d+, \d+, 0, ", r["name"]) for r in rows if r["scratch"] and any(h in r["name"] for h in kr.HOT)), \
        [(r["name"], r["scratch"]) for r in rows if r["scratch"]]


class _FakeGpu(object):
    """An oracle env behind the few VecCollisionAvoidanceEnv calls bench.verify_against_oracle makes: exercises the checker's
    own logic (which arenas, which actions at which step, the timed region's counters) without a GPU."""

    def __init__(self, A, N, scn, p, pool, warm, timed, corrupt=False):
        from tests import helpers as H
        from oracle import oracle as o
        self.A, self.N, self.o = A, N, o
        self.env = H.make_oracle(A, N, scn, p, seed=0)
        for s in range(warm + timed):
            if s == warm:
                self.before = self.env.get(o.FLD_ARENA_STATS).copy()
            self.env.step(pool[s % len(pool)], flags=o.F_STATS | o.F_OBS)
        if corrupt:
            x = self.env.get(o.FLD_POS_X)
            x[A - 1, N - 1] = np.nextafter(x[A - 1, N - 1], np.float32(100))       # one ulp in one agent of the batch
            self.env.set(o.FLD_POS_X, x)

    def get(self, field):
        from collision_avoidance_amd import _lib
        name = [k for k in dir(_lib) if k.startswith("FLD_") and getattr(_lib, k) == field][0]
        v = self.env.get(getattr(self.o, name))
        return (v - self.before) if name == "FLD_ARENA_STATS" else v

    def neighbor_lists(self):
        return self.env.get(self.o.FLD_NB_COUNT), self.env.get(self.o.FLD_NB_IDX)


def test_bench_verify_checker_logic():
    """bench.py --verify: the replayed block of arenas, the action of every step and the counters of the timed region line
    up with what the environment ran -- and a single ulp in a single agent is reported."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    A, N, warm, timed = 12, 16, 37, 21
    w = dict(n_arenas=A, n_agents=N, neighbor_dist=5.0, max_neighbors=10)
    p = scenarios.bench_params(N, 5.0, 10)
    pool = np.random.RandomState(5).uniform(-0.5, 0.5, (16, A, N)).astype(np.float32)
    args = argparse.Namespace(verify=12, mode="step", variant="walls")
    ok = bench.verify_against_oracle(_FakeGpu(A, N, "crowd", p, pool, warm, timed), args, w, p, "crowd", 0, pool, warm + timed, timed)
    assert ok["bit_exact"] and ok["arenas"] == 12 and ok["steps"] == warm + timed, ok
    args.verify = 5            # a block in the middle of the batch
    ok = bench.verify_against_oracle(_FakeGpu(A, N, "crowd", p, pool, warm, timed), args, w, p, "crowd", 0, pool, warm + timed, timed)
    assert ok["bit_exact"] and ok["arenas"] == 5 and ok["first_global_arena"] > 0, ok
    args.verify = 12
    bad = bench.verify_against_oracle(_FakeGpu(A, N, "crowd", p, pool, warm, timed, corrupt=True), args, w, p, "crowd", 0, pool,
                                      warm + timed, timed)
    assert not bad["bit_exact"] and bad["mismatch"].startswith("POS_X: 1 of"), bad
    off = bench.verify_against_oracle(_FakeGpu(A, N, "crowd", p, pool, warm, timed), args, w, p, "crowd", 0, pool, warm + timed + 1, timed)
    assert not off["bit_exact"]                                    # one step more or less is a mismatch too
