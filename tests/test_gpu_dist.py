"""The N > 1 path of bench.py with the HIP library under it, on the one-GPU box: two gloo ranks share the card
(CA_BENCH_BACKEND=gloo; the production backend is RCCL, one rank per GPU).  Started as a FRESH child process -- the
launcher (bench.py::launch_ranks -> torch.distributed.run) must never be reached by re-executing a process that has
already touched the GPU.  What is checked: the world really has two ranks, each reports its own agent-steps and
device, and the job statistics that came through the ONE all_gather equal the sum of two single-process runs that
play rank 0 and rank 1 (arena offsets 0 and 1024: the scenario RNG is keyed by the global arena id; the reference
replicates environments over workers the same way, run_rllib.py:108)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARM = 50, 10
INT_KEYS = ("agent_steps", "episodes", "collisions", "obst_collisions", "goals_reached", "obst_overflow")


def _bench(extra, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["CA_BENCH_MIN_WARM"] = "0"      # exactly --warmup steps: the runs below must start the timed region in the same state
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C2", "--steps", str(STEPS), "--warmup", str(WARM),
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_gloo_ranks_drive_the_hip_library():
    two = _bench(["--gpus", "2"], {"CA_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["world_size"] == 2 and two["scaling"] == "weak"
    per_rank = 1024 * 16 * STEPS
    assert [r["rank"] for r in two["ranks"]] == [0, 1]
    assert [r["agent_steps"] for r in two["ranks"]] == [per_rank, per_rank]
    assert two["stats"]["agent_steps"] == 2 * per_rank
    assert two["value"] > 0 and abs(two["value"] - 2 * 1024 * 16 / (two["ms_per_step"] * 1e-3)) <= 1e-6 * two["value"]
    singles = [_bench(["--as-rank", str(r)], {}) for r in (0, 1)]
    for k in INT_KEYS:
        assert two["stats"][k] == singles[0]["stats"][k] + singles[1]["stats"][k], (k, two["stats"], [s["stats"] for s in singles])
    assert two["stats"]["goals_reached"] > 0        # the crowd did something in 50 steps
    # every rank ran the HIP kernels: the launch geometry is that of the 1024 x 16 shard
    assert two["launch"] == singles[0]["launch"]
