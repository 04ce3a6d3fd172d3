"""The N > 1 path of bench.py with the HIP library under it, on the one-GPU box: two (and four) gloo ranks share the card
(CA_BENCH_BACKEND=gloo; the production backend is RCCL, one rank per GPU).  Started as a FRESH child process -- the
launcher (bench.py::launch_ranks -> torch.distributed.run) must never be reached by re-executing a process that has
already touched the GPU.  What is checked: the world really has two ranks, each reports its own agent-steps and
device, and the job statistics that came through the ONE all_gather equal the sum of two single-process runs that
play rank 0 and rank 1 (arena offsets 0 and 1024: the scenario RNG is keyed by the global arena id; the reference
replicates environments over workers the same way, run_rllib.py:108).  The pool allows at most six processes on the
card at once, so the widest rehearsal WITH the GPU is world 4 (test process + four ranks); world 8 is rehearsed on the
CPU (tests/test_dist_cpu.py::test_eight_ranks_gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARM = 50, 10
INT_KEYS = ("agent_steps", "episodes", "collisions", "obst_collisions", "goals_reached", "obst_overflow")


def _bench(extra, env_extra, workload=("--workload", "C2")):
    env = dict(os.environ)
    env.update(env_extra)
    env["CA_BENCH_MIN_WARM"] = "0"      # exactly --warmup steps: the runs below must start the timed region in the same state
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + list(workload) + ["--steps", str(STEPS), "--warmup", str(WARM),
                                                                              "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("world,arenas", [(2, 1024), (4, 128)])
def test_gloo_ranks_drive_the_hip_library(world, arenas):
    wl = ("--workload", "C2", "--arenas", str(arenas))
    job = _bench(["--gpus", str(world)], {"CA_BENCH_BACKEND": "gloo"}, wl)
    assert job["n_gpus"] == world and job["world_size"] == world and job["scaling"] == "weak"
    assert job["backend"] == "gloo" and job["collectives"]["data_path"] == 0
    per_rank = arenas * 16 * STEPS
    assert [r["rank"] for r in job["ranks"]] == list(range(world))
    assert [r["agent_steps"] for r in job["ranks"]] == [per_rank] * world      # counted in the kernels, per rank
    assert job["stats"]["agent_steps"] == world * per_rank
    assert job["value"] > 0 and abs(job["value"] - world * arenas * 16 / (job["ms_per_step"] * 1e-3)) <= 1e-6 * job["value"]
    singles = [_bench(["--as-rank", str(r)], {}, wl) for r in range(world)]
    for k in INT_KEYS:
        assert job["stats"][k] == sum(s["stats"][k] for s in singles), (k, job["stats"], [s["stats"] for s in singles])
    assert job["stats"]["goals_reached"] > 0        # the crowd did something in 50 steps
    # --verify under --gpus N: EVERY rank replayed arenas of its own global range through the oracle and its verdict came
    # through the one all_gather; the blocks differ by the ranks' arena offsets
    v = job["verified"]
    assert v["bit_exact"] and v["ranks_verified"] == world and v["steps"] == STEPS + WARM, v
    assert [r["verified"] for r in job["ranks"]] == [1] * world
    a0 = [r["verify_arena0"] for r in job["ranks"]]
    assert all(r * arenas <= a < (r + 1) * arenas for r, a in enumerate(a0)), a0
    assert [s["ranks"][0]["verify_arena0"] for s in singles] == a0     # a single process playing rank r checks the same block
    # the line explains itself per rank: each rank's own time for its K steps (before the closing barrier), its time including the
    # barrier (what the job's time is the maximum of), the warm-up it really ran, and the slowest-over-fastest ratio -- all from the
    # ONE all_gather (no further collective)
    for r in job["ranks"]:
        assert 0 < r["own_ns"] <= r["dt_ns"] and r["warmup_steps_run"] == WARM, r
        assert abs(r["agent_steps_per_s"] - r["agent_steps"] / r["own_ns"] * 1e9) <= 1e-6 * r["agent_steps_per_s"]
    assert abs(max(r["dt_ns"] for r in job["ranks"]) * 1e-6 / STEPS - job["ms_per_step"]) <= 1e-9 * job["ms_per_step"] + 1e-12
    rates = [r["agent_steps_per_s"] for r in job["ranks"]]
    assert 0 < job["efficiency"] <= 1 and abs(job["efficiency"] - min(rates) / max(rates)) < 1e-9
    assert job["collectives"]["job"].startswith("one all_gather")
    # every rank ran the HIP kernels: the launch geometry is that of its shard; the line names the code that ran
    assert job["launch"] == singles[0]["launch"]
    assert job["src_sha"] == singles[0]["src_sha"] == job["src_sha_on_disk"] != "unknown"


@pytest.mark.gpu
def test_rccl_carries_the_barriers_and_the_record_in_a_world_of_one():
    """The production backend itself ("nccl" = RCCL) on the one-GPU box: a world of one rank initialises the process group on
    its device, takes both timing barriers and the job's one all_gather (an int64 record on the GPU) through RCCL --
    the calls every rank of the 8-GPU job makes, which the gloo rehearsals above do not touch."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    wl = ("--workload", "C2", "--arenas", "256")
    env = {"CA_BENCH_FORCE_PG": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    job = _bench([], env, wl)
    assert job["backend"] == "nccl" and job["world_size"] == 1 and job["collectives"]["timing_barriers"] == 2
    assert len(job["ranks"]) == 1 and job["ranks"][0]["agent_steps"] == 256 * 16 * STEPS and job["ranks"][0]["device"] == 0
    assert job["ranks"][0]["verified"] == 1 and job["verified"]["bit_exact"]       # the verdict rode the RCCL all_gather too
    single = _bench([], {}, wl)
    for k in INT_KEYS:
        assert job["stats"][k] == single["stats"][k], (k, job["stats"], single["stats"])
