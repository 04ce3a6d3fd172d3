"""N > 1 path on CPU: two gloo ranks each own a shard of the arenas; the sharded run must equal the
single-process run arena by arena (scenario RNG keyed by global arena id) and the one collective
(all_gather of the statistics) must add up.  The compute stand-in here is the oracle -- the HIP
path has the same sharding test on the GPU (test_gpu_parity.py::test_sharding_invariance)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from collision_avoidance_amd import dist as cad
from collision_avoidance_amd import scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A_TOTAL, N, STEPS = 10, 16, 120


def test_shard_partition():
    for total in (1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            parts = [cad.shard(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(n for _, n in parts) == total
            for (o0, n0), (o1, _) in zip(parts, parts[1:]):
                assert o1 == o0 + n0
            assert max(n for _, n in parts) - min(n for _, n in parts) <= 1
    assert cad.weak_shard(4096, 3) == (3 * 4096, 4096)
    with pytest.raises(ValueError):
        cad.shard(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_oracle(offset, n_local):
    from oracle import oracle as o
    from tests import helpers as H
    p = scenarios.bench_params(N, 1.5, 5)
    env = H.make_oracle(n_local, N, "crowd", p, seed=21, arena_offset=offset)
    env.rollout(STEPS, flags=o.F_STATS)
    return env.get(o.FLD_POS_X), env.get(o.FLD_GOAL_X), env.stats()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offset, n_local = cad.shard(A_TOTAL, rank, world)
    px, gx, st = _run_oracle(offset, n_local)
    per_rank, total = cad.gather_stats(st, extra={"device": 10 + rank})     # ONE all_gather: counters, reward bits, extras
    assert [d["device"] for d in per_rank] == [10 + r for r in range(world)]
    assert per_rank[rank]["sum_reward"] == st["sum_reward"]                 # the fp64 sum travels as its bits
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), px=px, gx=gx, offset=offset,
             total=np.array([total[k] for k in cad.STAT_KEYS]),
             mine=np.array([per_rank[rank][k] for k in cad.STAT_KEYS]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    px, gx, st = _run_oracle(0, A_TOTAL)                     # the whole job in one process
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    assert [int(p["offset"]) for p in parts] == [0, 5]
    np.testing.assert_array_equal(np.concatenate([p["px"] for p in parts]), px)
    np.testing.assert_array_equal(np.concatenate([p["gx"] for p in parts]), gx)
    whole = np.array([st[k] for k in cad.STAT_KEYS])
    for p in parts:                                            # every rank holds the job totals
        np.testing.assert_array_equal(p["total"], whole)
    np.testing.assert_array_equal(parts[0]["mine"] + parts[1]["mine"], whole)
    assert whole[0] == A_TOTAL * N * STEPS


def _worker8(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offset, n_local = cad.weak_shard(2, rank)                    # weak scaling like bench.py: 2 arenas per rank
    px, gx, st = _run_oracle(offset, n_local)
    # the job's ONE collective carries the statistics, the device and the rank's own time (bench.py): max over the records
    per_rank, total = cad.gather_stats(st, extra={"device": rank, "dt_ns": 1000 + 7 * rank})
    assert max(d["dt_ns"] for d in per_rank) == 1000 + 7 * (world - 1)
    assert [d["device"] for d in per_rank] == list(range(world))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), px=px, total=np.array([total[k] for k in cad.STAT_KEYS]))
    dist.destroy_process_group()


def test_eight_ranks_gloo(tmp_path):
    """World size 8 (BASELINE config C4's shape: one shard per GPU, weak scaling) rehearsed on the CPU: the shards
    are the slices of the single-process run, and every rank ends with the job totals after one all_gather."""
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    px, gx, st = _run_oracle(0, 16)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(8)]
    np.testing.assert_array_equal(np.concatenate([p["px"] for p in parts]), px)
    whole = np.array([st[k] for k in cad.STAT_KEYS])
    for p in parts:
        np.testing.assert_array_equal(p["total"], whole)
    assert whole[0] == 16 * N * STEPS


def test_gather_stats_single_process():
    st = dict(agent_steps=10, episodes=1, collisions=2, obst_collisions=0, goals_reached=3,
              obst_overflow=0, sum_reward=1.5)
    per_rank, total = cad.gather_stats(st)
    assert per_rank == [st] and total == st
    per_rank, total = cad.gather_stats(st, extra={"device": 3})
    assert per_rank[0]["device"] == 3 and total == st
