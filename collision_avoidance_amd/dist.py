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


def gather_stats(stats, device=None, group=None):
    """all_gather of the integer statistics (+ sum_reward) of every rank; returns the per-rank list
    and the job totals.  Works without an initialised process group (single process)."""
    import torch
    import torch.distributed as dist
    ints = torch.tensor([int(stats[k]) for k in STAT_KEYS], dtype=torch.int64, device=device)
    rew = torch.tensor([float(stats.get("sum_reward", 0.0))], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        per_rank = [dict(stats)]
    else:
        world = dist.get_world_size(group)
        gi = [torch.zeros_like(ints) for _ in range(world)]
        gr = [torch.zeros_like(rew) for _ in range(world)]
        dist.all_gather(gi, ints, group=group)
        dist.all_gather(gr, rew, group=group)
        per_rank = []
        for a, b in zip(gi, gr):
            d = dict(zip(STAT_KEYS, (int(v) for v in a.tolist())))
            d["sum_reward"] = float(b.item())
            per_rank.append(d)
    total = {k: sum(d[k] for d in per_rank) for k in STAT_KEYS}
    total["sum_reward"] = sum(d["sum_reward"] for d in per_rank)
    return per_rank, total
