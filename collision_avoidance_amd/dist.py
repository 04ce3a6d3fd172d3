"""Multi-GPU plumbing: one process per GPU, arenas sharded, ONE collective.

Arenas never interact (every reference env owns its own simulator, collision_avoidence_env.py:62;
RLlib replicates envs across worker processes, run_rllib.py:108), so the data path needs no
exchange at all.  Each rank owns a contiguous range of global arena ids -- the scenario RNG is keyed
by the global id, so results do not depend on the number of ranks -- and the only collective of a
job is an all_gather of the per-rank statistics (torch.distributed: "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).
"""
import os

STAT_KEYS = ("agent_steps", "episodes", "collisions", "obst_collisions", "goals_reached", "obst_overflow")


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def shard(n_total, rank, world):
    """Contiguous block partition of n_total arenas: returns (arena_offset, n_local)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(n_total, world)
    n_local = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, n_local


def weak_shard(n_per_rank, rank):
    """Weak scaling (BASELINE config C4: 4096 arenas per GPU): rank r owns [r*n, (r+1)*n)."""
    return rank * n_per_rank, n_per_rank


def gather_stats(stats, device=None, group=None, extra=None):
    """THE collective of a job: one all_gather of a fixed per-rank record -- the integer statistics, the bits of
    the fp64 sum_reward and any `extra` integers (e.g. the device a rank ran on) -- returns the per-rank list and
    the job totals.  Works without an initialised process group (single process)."""
    import struct
    import torch
    import torch.distributed as dist
    extra = dict(extra or {})
    rew_bits = struct.unpack("<q", struct.pack("<d", float(stats.get("sum_reward", 0.0))))[0]
    rec = torch.tensor([int(stats[k]) for k in STAT_KEYS] + [rew_bits] + [int(v) for v in extra.values()],
                       dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        rows = [rec.tolist()]
    else:
        world = dist.get_world_size(group)
        got = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(got, rec, group=group)
        rows = [g.tolist() for g in got]
    per_rank = []
    nk = len(STAT_KEYS)
    for row in rows:
        d = dict(zip(STAT_KEYS, (int(v) for v in row[:nk])))
        d["sum_reward"] = struct.unpack("<d", struct.pack("<q", int(row[nk])))[0]
        for k, v in zip(extra.keys(), row[nk + 1:]):
            d[k] = int(v)
        per_rank.append(d)
    total = {k: sum(d[k] for d in per_rank) for k in STAT_KEYS}
    total["sum_reward"] = sum(d["sum_reward"] for d in per_rank)
    return per_rank, total
