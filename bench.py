#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the batched collision-avoidance step on MI355X.

One "step" = one pass of the hot path over every arena of the workload: the full environment
step of the reference (collision_avoidence_env.py:367-416: action -> preferred velocity ->
ORCA doStep -> reward -> done test -> 16-ray laser observation) for 4096 arenas x 64 agents per
GPU (BASELINE.json configs[2], "C3": random start/goal crowd, neighborDist 5, maxNeighbors 10),
state and actions resident in HBM.  value = agents advanced per second over all GPUs.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2|C5] [--mode step|orca]
                  [--no-cpu-baseline]
N > 1: launched by torch.distributed.run, one rank per GPU; arenas are sharded (no data-path
collective: arenas never interact), one RCCL all_gather of the per-rank statistics at the end.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# algorithmic bytes per agent-step
BYTES_STEP_KERNEL_FULL = 60   # read pos 8 vel 8 goal 8 done 4 action 4; write pos 8 vel 8 done 4 stat 4 reward 4
BYTES_STEP_KERNEL_ORCA = 52
BYTES_OBS_KERNEL = 256        # the 64-float observation row


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C5"])
    ap.add_argument("--mode", default="step", choices=["step", "orca"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(workload, mode, seconds):
    """The CPU oracle (a C++ restatement of the same path) on a bounded sample of the same workload: arenas of the
    same scenario stepped for about `seconds`, first on one core, then on all cores of this process's share of
    the host (arenas dealt to threads).  `value` is the multi-core rate, `cores` the threads used."""
    from collision_avoidance_amd import scenarios
    from oracle import oracle as o
    from tests import helpers as H
    w = scenarios.BENCH_CONFIGS[workload]
    N = w["n_agents"]
    try:
        cores = max(1, min(32, len(os.sched_getaffinity(0))))
    except Exception:
        cores = max(1, min(32, os.cpu_count() or 1))
    rng = np.random.RandomState(0)

    def run(A, threads, budget):
        p = scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"])
        env = H.make_oracle(A, N, "crowd", p, seed=0)
        acts = rng.uniform(-0.5, 0.5, (8, A, N)).astype(np.float32)
        flags = o.F_OBS if mode == "step" else 0
        for s in range(3):
            env.step_mt(acts[s] if mode == "step" else None, flags=flags, n_threads=threads)
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < budget:
            for s in range(5):
                env.step_mt(acts[(steps + s) % 8] if mode == "step" else None, flags=flags, n_threads=threads)
            steps += 5
        dt = time.perf_counter() - t0
        return A * N * steps / dt, steps, dt

    A1 = max(1, min(64, 4096 // N))
    v1, s1, d1 = run(A1, 1, seconds * 0.4)
    Am = max(cores, min(16 * cores, (4096 // N) * cores // 4 or cores))
    vm, sm, dm = run(Am, cores, seconds * 0.6)
    return {"value": vm, "unit": "agent-steps/s", "cores": cores, "kind": "port", "single_core_value": v1,
            "sample": "%d arenas x %d agents x %d steps on %d threads (%.1f s) and %d arenas x %d steps on one (%.1f s), "
                      "same workload (%s mode), oracle/ca_oracle.cpp -O2" % (Am, N, sm, cores, dm, A1, s1, d1, mode)}


def main():
    args = parse()
    from collision_avoidance_amd import dist as cad
    rank, world, local = cad.rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # rehearsal switches (not used by the driver): CA_BENCH_BACKEND=gloo lets several ranks share one
    # GPU on a 1-GPU box; the production path is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("CA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    from collision_avoidance_amd import build as _b
    if rank == 0 and not os.path.exists(_b.LIB_PATH):  # normally prebuilt by __graft_entry__.build()
        _b.build()
    if dist is not None:
        dist.barrier()
    from collision_avoidance_amd import scenarios
    from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
    from collision_avoidance_amd import _lib

    w = scenarios.BENCH_CONFIGS[args.workload]
    A, N = w["n_arenas"], w["n_agents"]
    p = scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"])
    arena_offset, _ = cad.weak_shard(A, rank)  # weak scaling: every GPU owns A arenas of the global range
    env = VecCollisionAvoidanceEnv(A, N, scenario="crowd", params=p, device=local, seed=0,
                                   arena_offset=arena_offset, use_torch=True)
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    pool = (torch.rand((16, A, N), device="cuda", generator=gen) - 0.5)  # actions in [-0.5, 0.5] rad
    full = args.mode == "step"

    def one_step(i):  # the production call: one ca_step (neighbours -> ORCA solve -> observation)
        if full:
            env._call("ca_step", env.h, pool[i % 16].data_ptr(), _lib.F_STATS | _lib.F_OBS)
        else:
            env._call("ca_orca_step", env.h, _lib.F_STATS)

    for i in range(args.warmup):
        one_step(i)
    # the kernel launches of every 8th step of the timed region are bracketed by HIP events on the
    # stream they run on (recorded inside the library, which is where the launches are issued);
    # sampling keeps the event records from stretching the timed region (every launch: +5 % wall)
    env.profile(8)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ktimes = env.profile_read()
    env.profile(0)
    st = env.stats()

    # the single collective of the job: per-rank statistics (RCCL all_gather over xGMI when N > 1)
    per_rank_stats, total_stats = cad.gather_stats(st, device=coll_dev)
    if rank == 0:
        agents = A * N
        value = world * agents * args.steps / dt
        kbytes_of = {"nbr_kernel": 0, "step_kernel": BYTES_STEP_KERNEL_FULL if full else BYTES_STEP_KERNEL_ORCA,
                     "obs_kernel": BYTES_OBS_KERNEL}
        kms_of = {k: v[1] for k, v in ktimes.items() if v[0] > 0 and k in kbytes_of}
        # dominant kernel: the longest average launch among the kernels that own algorithmic HBM bytes
        # (nbr_kernel has none); launches within 3 % of the longest count as a tie, decided by the bytes moved
        longest = max(v for k, v in kms_of.items() if kbytes_of[k] > 0)
        dom = max((k for k in kms_of if kbytes_of[k] > 0 and kms_of[k] >= 0.97 * longest), key=lambda k: kbytes_of[k])
        kms, kbytes = kms_of[dom], kbytes_of[dom]
        achieved = agents * kbytes / (kms * 1e-3) / 1e9
        per_kernel = {k: {"algorithmic_GB_per_s": agents * kbytes_of[k] / (v * 1e-3) / 1e9,
                          "frac": agents * kbytes_of[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS}
                      for k, v in kms_of.items() if kbytes_of[k] > 0}
        traffic = None
        try:  # HBM bytes per launch from this round's committed rocprofv3 --pmc passes of this command
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_final_hbm_traffic_pmc.json")))
            if args.workload == "C3" and full:
                traffic = tj["kernels"][dom]["hbm_bytes_high"]
        except Exception:
            traffic = None
        # vector-ALU view (SURVEY 8d asks for it next to the HBM fraction): wave-level VALU instructions per launch
        # from this round's committed SQ counter pass of this command x 64 lanes, over the live kernel times,
        # against the chip's issue peak of 256 CUs x 128 lanes x 2.4 GHz lane-instructions/s (4 SIMD-32 per CU: a
        # wave64 instruction takes 2 cycles when two waves alternate, 4 for a wave alone; = 157 TFLOP/s for FMAs)
        valu = None
        try:
            if args.workload == "C3" and full:
                cnt, cur = {}, None
                for line in open(os.path.join(ROOT, "profiles", "r01_final_sq_pmc.txt")):
                    if "kernel" in line and "launches" in line:
                        cur = "step_kernel" if "step_kernel" in line else ("obs_kernel" if "obs_kernel" in line else None)
                    elif cur and line.split() and line.split()[0] == "SQ_INSTS_VALU":
                        cnt[cur] = float(line.split()[1])
                lane_ops = 64.0 * sum(cnt[k] for k in kms_of)
                peak = 256 * 128 * 2.4e9
                valu = {"wave_insts_per_step": {k: cnt[k] for k in kms_of}, "lane_ops_per_s": lane_ops / (sum(kms_of.values()) * 1e-3),
                        "peak_lane_ops_per_s": peak, "frac": lane_ops / (sum(kms_of.values()) * 1e-3) / peak,
                        "per_kernel_frac": {k: 64.0 * cnt[k] / (kms_of[k] * 1e-3) / peak for k in kms_of},
                        "source": "profiles/r01_final_sq_pmc.txt (SQ_INSTS_VALU per launch) over the live kernel times"}
        except Exception:
            valu = None
        out = {
            "metric": "agent-steps/sec (whole node), %d arenas x %d agents per GPU" % (A, N),
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d arenas x %d agents per GPU, random start/goal crowd (ALAN recipe), "
                                   "neighborDist %.1f, maxNeighbors %d, %s" %
                                   (args.workload, A, N, w["neighbor_dist"], w["max_neighbors"],
                                    "full env step (action -> ORCA -> reward/done -> laser obs)" if full
                                    else "ORCA-only step (no observation)"),
                       "mode": args.mode, "sharding": "arenas, %d per GPU" % A},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_agent": kbytes, "kernel_ms": kms, "per_kernel": per_kernel},
            "valu": valu,
            "kernels_ms": dict({k: round(v, 5) for k, v in kms_of.items()}, sum=round(sum(kms_of.values()), 5),
                               wall_per_step=dt / args.steps * 1e3),
            "full_step_algorithmic": {"bytes_per_agent": 316 if full else 52,
                                      "GB_per_s": agents * (316 if full else 52) / (dt / args.steps) / 1e9},
            "stats": {k: total_stats[k] for k in cad.STAT_KEYS},
            "launch": env.launch_info(),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.mode, args.cpu_seconds)
        print(json.dumps(out))
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
