"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY section 5 "sanitizers"; CPU build only -- GPU
sanitizers are not available on this pool).  `make -C oracle asan` builds libca_oracle_asan.so from the same source; a child
interpreter started with LD_PRELOAD=libasan loads it (CA_ORACLE_SANITIZED=1) and replays, through the ordinary test
functions, one reference-generated env fixture (the episode that ends, retargets and resets), one ALAN online fixture, one
finished run_sim episode, the threaded rollout, and a 200-step C3-shaped crowd against its boundary walls with observation,
statistics and auto-reset -- the oracle paths bench.py's cpu_baseline and --verify legs drive.  Any sanitizer report fails the
test (halt_on_error, and the log is scanned as well)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CROWD = r'''
import numpy as np
from oracle import oracle as o
from tests import helpers as H
from collision_avoidance_amd import scenarios
assert o._LIB_PATH.endswith("libca_oracle_asan.so"), o._LIB_PATH
p = scenarios.bench_params(64, 5.0, 10)
env = H.make_oracle(3, 64, "crowd", p, seed=0)                       # C3's shape: 64 agents, K = 10, four boundary walls
rng = np.random.RandomState(0)
acts = rng.uniform(-0.5, 0.5, (8, 3, 64)).astype(np.float32)
env.rollout_mt(100, acts, flags=o.F_STATS | o.F_OBS, n_threads=3)
for s in range(100):
    env.step(acts[s % 8], flags=o.F_STATS | o.F_OBS | o.F_AUTORESET)
env.orca_step(flags=o.F_OBS | o.F_STATS)
st = env.stats()
assert st["agent_steps"] == 3 * 64 * 201, st
assert np.isfinite(env.get(o.FLD_OBS)).all() and np.isfinite(env.get(o.FLD_POS_X)).all()
big = H.make_oracle(1, 512, "crowd", scenarios.bench_params(512, 5.0, 10), seed=1)      # C5's shape (grid / wide lists)
big.rollout_mt(5, None, flags=o.F_STATS, n_threads=1)
print("CROWD_OK", st["collisions"], st["goals_reached"])
'''


def _sanitizer_env():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, CA_ORACLE_SANITIZED="1", LD_PRELOAD=libasan, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98")
    return env


def _run(cmd, env):
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    log = r.stdout
    assert "AddressSanitizer" not in log and "runtime error:" not in log and "LeakSanitizer" not in log, log[-6000:]
    assert r.returncode == 0, (r.returncode, log[-6000:])
    return log


def test_sanitized_oracle_builds_and_replays_goldens_and_a_crowd():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = _sanitizer_env()
    log = _run([sys.executable, "-c", CROWD], env)
    assert "CROWD_OK" in log, log[-2000:]
    log = _run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                "tests/test_oracle_golden.py::test_env_loop_f64_matches_reference",
                "tests/test_oracle_golden.py::test_alan_scenarios_match_reference",
                "tests/test_oracle_alan.py::test_online_step_matches_reference",
                "tests/test_oracle_alan.py::test_finished_episode_matches_reference",
                "tests/test_oracle_orca.py"], env)
    assert " passed" in log and "failed" not in log, log[-2000:]
