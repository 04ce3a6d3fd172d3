#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the batched collision-avoidance step on MI355X.

One "step" = one pass of the hot path over every arena of the workload: the full environment
step of the reference (collision_avoidence_env.py:367-416: action -> preferred velocity ->
ORCA doStep -> reward -> done test -> 16-ray laser observation) for 4096 arenas x 64 agents per
GPU (BASELINE.json configs[2], "C3": random start/goal crowd, neighborDist 5, maxNeighbors 10),
state and actions resident in HBM.  value = agents advanced per second over all GPUs.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2|C5] [--mode step|orca]
                  [--variant walls|free] [--starts overlap|separated] [--no-cpu-baseline]

N > 1: one rank per GPU (torch.distributed.run; when this script is started directly with --gpus N it
starts the N ranks itself, as a child process, before anything touches the GPU, and relays rank 0's line);
arenas are sharded (no data-path collective: arenas never interact), ONE RCCL all_gather of the per-rank
record at the end (statistics + the rank's time for the timed region, so the maximum over ranks needs no collective of
its own); the only other synchronisation is the barrier on either side of the timed region that the benchmark contract asks for.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
VALU_PEAK_LANE_OPS = 256 * 128 * 2.4e9  # 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz: the rate behind the 157 TFLOP/s fp32 peak,
#                                         which only packed FMAs reach (profiles/r02_valu_issue_rates_microbench.txt)
# what a SIMD really issues for the instruction mix of these kernels: ~4.2 cycles for most vector instructions, ~2.7 for
# plain add / mul / and / mov, 8.3 for rcp / sqrt (same file); the issue-bound estimate prices each instruction class
# from the class counters of the committed profile (4.0 per instruction if a profile has none)
ISSUE_CYCLES_PER_VALU = 4.0
N_SIMDS, CLOCK_HZ = 1024, 2.4e9
# algorithmic bytes per agent-step
BYTES_STEP_KERNEL_FULL = 60   # read pos 8 vel 8 goal 8 done 4 action 4; write pos 8 vel 8 done 4 stat 4 reward 4
BYTES_STEP_KERNEL_ORCA = 52
BYTES_OBS_KERNEL = 256        # the 64-float observation row
# clocks settle: the warm-up runs at least this long whatever --warmup says (CA_BENCH_MIN_WARM: diagnostic override)
MIN_WARM_SECONDS = float(os.environ.get("CA_BENCH_MIN_WARM", "1.0"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C5", "A16", "A50", "A100", "M128", "M180"])
    ap.add_argument("--mode", default="step", choices=["step", "orca", "alan"],
                    help="step: full env step (actions in, observation out); orca: ORCA-only rollout; alan: the ALAN online-learning "
                         "rollout of ALAN_true.py:106-123 (softmax draw -> ORCA step -> bandit update per agent and step, no observation)")
    ap.add_argument("--variant", default="walls", choices=["walls", "free"],
                    help="SURVEY 8d: A = with the boundary polygon (default), B = obstacle-free")
    ap.add_argument("--starts", default="overlap", choices=["overlap", "separated"],
                    help="SURVEY 8d: uniform starts (overlaps allowed, default) or rejection-sampled non-overlapping starts")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--rollout-chunk", type=int, default=50,
                    help="--mode orca: steps per ca_rollout call (the ORCA-only policy rollout of env.py:570-573 / ALAN:106-123); "
                         "1 = one ca_orca_step call per step")
    ap.add_argument("--arenas", type=int, default=None, help="diagnostic: override the workload's arena count (batch-size studies)")
    ap.add_argument("--verify", type=int, default=8,
                    help="after the timed region (outside it): replay this many arenas of the rank through the CPU oracle for EVERY step "
                         "the GPU ran (warm-up + timed) with the same actions and compare the state bit for bit; 0 = off")
    ap.add_argument("--as-rank", type=int, default=None,
                    help="rehearsal (tests): a single process plays rank R of a larger job -- arena offset R * arenas, action seed of rank R")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks with torch.distributed.run as a CHILD
    process (nothing in this process has touched the GPU), relay its output as it comes (a hung rank shows what it
    printed so far) and exit with its code.  --standalone lets the launcher's own c10d store pick a free port (no
    pre-picked port that another process could take between the probe and the bind)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    line = None
    for ln in child.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
            sys.stderr.flush()
    rc = child.wait()
    if rc != 0 or line is None:
        raise SystemExit(rc or 1)
    print(line)
    raise SystemExit(0)


def host_cores():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box shows all
    256 hardware threads of the host but gives a one-GPU job a share of them: threads beyond the quota only time-slice)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(path).read().split()
            if path.endswith("cpu.max"):
                if t[0] != "max":
                    n = min(n, max(1, int(int(t[0]) / int(t[1]) + 0.5)))
            else:
                q = int(t[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, min(n, 64))


def cpu_baseline(workload, mode, seconds, variant="walls", starts="overlap"):
    """The CPU oracle (a C++ restatement of the same path) on a bounded sample of the same workload: arenas of the
    same scenario stepped for about `seconds`, first on one core, then on all cores of this process's share of
    the host: a pool of threads, each stepping its own block of arenas through the whole sample without a per-step
    join (oracle/ca_oracle.cpp orc_env_rollout_mt).  `value` is the multi-core rate, `cores` the threads used."""
    import numpy as np
    from collision_avoidance_amd import scenarios
    from oracle import oracle as o
    from tests import helpers as H
    w = scenarios.BENCH_CONFIGS[workload]
    N = w["n_agents"]
    cores = host_cores()
    rng = np.random.RandomState(0)
    scn = w.get("scenario", "crowd") if starts == "overlap" else "crowd_separated"
    flags = o.F_OBS if mode == "step" else 0
    if mode == "alan":   # the oracle's ALAN step is serial: one core, one batch of arenas, a bounded sample
        from collision_avoidance_amd import alan as _alan
        p = scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"])
        p.update(reward_scale=0.6)
        A1 = max(1, min(64, 2048 // N))
        env = H.make_oracle(A1, N, scn, p, seed=0)
        env.alan_configure(_alan.DEFAULT_ACTIONS)
        for _ in range(3):
            env.alan_step(flags=o.F_STATS)
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                env.alan_step(flags=o.F_STATS)
            steps += 10
        dt = time.perf_counter() - t0
        return {"value": A1 * N * steps / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port", "single_core_value": A1 * N * steps / dt,
                "sample": "%d arenas x %d agents x %d ALAN online steps on one thread (%.1f s), oracle/ca_oracle.cpp -O2 "
                          "(its ALAN step has no thread pool)" % (A1, N, steps, dt)}

    def run(A, threads, budget):
        p = scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"])
        env = H.make_oracle(A, N, scn, p, seed=0, polys=None if variant == "walls" else [])
        acts = rng.uniform(-0.5, 0.5, (8, A, N)).astype(np.float32) if mode == "step" else None
        env.rollout_mt(2, acts, flags=flags, n_threads=threads)          # warm the caches, size the sample
        t0 = time.perf_counter()
        env.rollout_mt(4, acts, flags=flags, n_threads=threads)
        per_step = max(1e-6, (time.perf_counter() - t0) / 4)
        steps = int(max(8, min(4000, budget / per_step)))
        t0 = time.perf_counter()
        env.rollout_mt(steps, acts, flags=flags, n_threads=threads)
        dt = time.perf_counter() - t0
        return A * N * steps / dt, steps, dt

    A1 = max(1, min(64, 4096 // N))
    v1, s1, d1 = run(A1, 1, seconds * 0.4)
    Am = cores * max(1, min(16, 1024 // N))    # every thread owns a block of whole arenas
    vm, sm, dm = run(Am, cores, seconds * 0.6)
    return {"value": vm, "unit": "agent-steps/s", "cores": cores, "kind": "port", "single_core_value": v1,
            "sample": "thread pool: %d arenas x %d agents x %d steps on %d threads, every thread steps its own block of arenas "
                      "through the whole sample (%.1f s); and %d arenas x %d steps on one thread (%.1f s); same workload "
                      "(%s mode, %s, %s starts), oracle/ca_oracle.cpp -O2"
                      % (Am, N, sm, cores, dm, A1, s1, d1, mode, variant, starts)}


def verify_against_oracle(env, args, w, p, scn, arena_offset, pool, total_steps, timed_steps):
    """The parity check of THIS run, on the state the timed region really ran in: `k` arenas of this rank are replayed
    through the CPU oracle (the checker -- never the thing measured) from the initial scenario through all `total_steps`
    steps the GPU has executed (warm-up included) with the same action pool, and every state field is compared with the
    GPU's bit for bit; the collision counters of the timed region too.  Scenario draws, re-goals and ALAN draws are
    counter-based and keyed by the GLOBAL arena id, so a block of arenas replays on its own."""
    import numpy as np
    from collision_avoidance_amd import _lib
    from oracle import oracle as o
    from tests import helpers as H
    t0 = time.perf_counter()
    A, N = env.A, env.N
    k = max(1, min(args.verify, A, max(2, 2048 // N)))
    start = (1234 + 37 * (arena_offset // max(1, A))) % (A - k + 1)
    orc = H.make_oracle(k, N, scn, p, seed=0, arena_offset=arena_offset + start,
                        polys=None if args.variant == "walls" else [])
    threads = max(1, min(k, host_cores()))
    warm = total_steps - timed_steps

    def advance(first, n):            # steps [first, first + n) of the run
        if n <= 0:
            return
        if args.mode == "step":
            acts = np.roll(pool[:, start:start + k], -(first % pool.shape[0]), axis=0)    # this block's actions, step `first` first
            orc.rollout_mt(n - 1, acts, flags=o.F_STATS, n_threads=threads)        # the observation feeds nothing back: only
            orc.step(acts[(n - 1) % acts.shape[0]], flags=o.F_STATS | o.F_OBS)       # the last step's is computed (and compared)
        elif args.mode == "orca":
            orc.rollout_mt(n, None, flags=o.F_STATS, n_threads=threads)
        else:
            for _ in range(n):
                orc.alan_step(flags=o.F_STATS)
    if args.mode == "alan":
        from collision_avoidance_amd import alan as _alan
        orc.alan_configure(_alan.DEFAULT_ACTIONS)
    advance(0, warm)
    before = orc.get(o.FLD_ARENA_STATS).copy()
    advance(warm, timed_steps)
    names = ["POS_X", "POS_Y", "VEL_X", "VEL_Y", "PREF_X", "PREF_Y", "GOAL_X", "GOAL_Y", "GOAL2_X", "GOAL2_Y", "AGENT_DONE",
             "ARRIVE_STEP", "STEP_COUNT", "ARENA_DONE", "EPISODE", "REGOAL_COUNT", "NB_COUNT", "OBST_COUNT"]
    if args.mode == "step":
        names += ["REWARD", "OBS"]
    if args.mode == "alan":
        names += ["ALAN_ACTION", "ALAN_WEIGHTS", "ALAN_TIMES"]
    mismatch = None

    def bits(a):
        a = np.ascontiguousarray(a)
        return a.view({4: np.uint32, 8: np.uint64, 2: np.uint16, 1: np.uint8}[a.dtype.itemsize])
    for name in names:
        g = env.get(getattr(_lib, "FLD_" + name))[start:start + k]
        c = orc.get(getattr(o, "FLD_" + name))
        if g.shape != c.shape or not np.array_equal(bits(g), bits(c)):
            bad = np.argwhere(bits(g) != bits(c)) if g.shape == c.shape else []
            mismatch = "%s: %d of %d entries differ%s" % (name, len(bad), g.size, (
                "; first at %s: gpu %r oracle %r" % (tuple(bad[0]), g[tuple(bad[0])], c[tuple(bad[0])])) if len(bad) else "")
            break
    if mismatch is None:              # the neighbour lists of the last step (entries beyond the count are not defined)
        gc, gi = env.neighbor_lists()
        oc, oi = orc.get(o.FLD_NB_COUNT), orc.get(o.FLD_NB_IDX)
        m = np.arange(oi.shape[2])[None, None, :] < oc[:, :, None]
        if not np.array_equal(np.where(m, gi[start:start + k], -1), np.where(m, oi, -1)):
            mismatch = "NB_IDX differs"
    if mismatch is None:              # the counters of the TIMED region (bench.py resets the GPU's after the warm-up)
        gs = env.get(_lib.FLD_ARENA_STATS)[start:start + k].astype(np.int64)
        cs = (orc.get(o.FLD_ARENA_STATS).astype(np.int64) - before.astype(np.int64))
        for col, what in ((1, "collisions"), (2, "obst_collisions"), (3, "goals_reached"), (4, "obst_overflow")):
            if not np.array_equal(gs[:, col], cs[:, col]):
                mismatch = "per-arena %s of the timed region: gpu %s oracle %s" % (what, gs[:, col].tolist(), cs[:, col].tolist())
                break
    return {"arenas": k, "first_global_arena": arena_offset + start, "steps": total_steps, "timed_steps": timed_steps,
            "fields": len(names) + 2, "bit_exact": mismatch is None, "mismatch": mismatch,
            "seconds": round(time.perf_counter() - t0, 2),
            "how": "CPU oracle (oracle/ca_oracle.cpp) stepped from the initial scenario with the same action pool; state, lists, "
                   "reward / observation of the last step and the timed region's per-arena counters compared as bit patterns"}


def counters_for(workload, mode, variant, starts, kernels):
    """HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and SQ_INSTS_VALU of this command from the newest committed
    rocprofv3 --pmc summary (profiles/*_counters_<workload>_<mode>.json, written by tools/counters.py) -- quoted
    ONLY if that summary was taken from the kernel sources the loaded library was built from (source hash match);
    otherwise null with the reason, so a changed kernel can never carry stale counter data."""
    import glob
    from collision_avoidance_amd import build as _b
    sha = _b.loaded_sha()   # the hash compiled into the library that ran
    tag = "%s_%s" % (workload, mode) + ("" if variant == "walls" else "_free") + ("" if starts == "overlap" else "_separated")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters_%s.json" % tag)))
    if not files:
        return None, "no counter profile committed for %s" % tag
    for f in reversed(files):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("src_sha") == sha:
            return j, os.path.relpath(f, ROOT)
    return None, "newest counter profile (%s) was taken from other kernel sources than this build (%s)" % (
        os.path.basename(files[-1]), sha)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    import numpy as np  # noqa: F401
    import torch
    from collision_avoidance_amd import dist as cad
    rank, world, local = cad.rank_world()
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # rehearsal switches (not used by the driver): CA_BENCH_BACKEND=gloo lets several ranks share one
    # GPU on a 1-GPU box; the production path is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("CA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist = None
    # CA_BENCH_FORCE_PG=1 (rehearsal, tests/test_gpu_dist.py): a world of one still initialises the process group, so the
    # one-GPU box takes the barriers and the all_gather through RCCL exactly as a rank of an 8-GPU job does
    if world > 1 or os.environ.get("CA_BENCH_FORCE_PG") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    from collision_avoidance_amd import build as _b
    if local == 0 and not os.path.exists(_b.LIB_PATH):  # normally prebuilt by __graft_entry__.build()
        _b.build()
    oracle_error = None
    if local == 0 and args.verify > 0:
        # the checker's library is built by ONE process per node, before the barrier: N ranks calling make at once could
        # write libca_oracle.so while another rank maps it (it too is normally prebuilt by __graft_entry__.build())
        try:
            from oracle import oracle as _o
            _o.build()
        except Exception as e:      # a box without make / g++: the measurement still stands, the verdict says why it is missing
            oracle_error = "%s: %s" % (type(e).__name__, e)
    if dist is not None:
        dist.barrier()
    from collision_avoidance_amd import scenarios
    from collision_avoidance_amd.vec_env import VecCollisionAvoidanceEnv
    from collision_avoidance_amd import _lib

    w = scenarios.BENCH_CONFIGS[args.workload]
    A, N = (args.arenas or w["n_arenas"]), w["n_agents"]
    p = scenarios.bench_params(N, w["neighbor_dist"], w["max_neighbors"])
    job_rank = rank if args.as_rank is None else args.as_rank
    arena_offset, _ = cad.weak_shard(A, job_rank)  # weak scaling: every GPU owns A arenas of the global range
    scn = w.get("scenario", "crowd") if args.starts == "overlap" else "crowd_separated"
    if args.mode == "alan":
        p.update(reward_scale=0.6)    # ALAN:47 gamma
    env = VecCollisionAvoidanceEnv(A, N, scenario=scn, params=p, device=local, seed=0,
                                   arena_offset=arena_offset, use_torch=True,
                                   obstacles="scenario" if args.variant == "walls" else [])
    gen = torch.Generator(device="cuda").manual_seed(1234 + job_rank)
    pool = (torch.rand((16, A, N), device="cuda", generator=gen) - 0.5)  # actions in [-0.5, 0.5] rad
    full = args.mode == "step"
    alan_mode = args.mode == "alan"
    if alan_mode:
        from collision_avoidance_amd import alan as _alan
        env.alan_configure(_alan.DEFAULT_ACTIONS)

    chunk = 1 if full else max(1, args.rollout_chunk)
    if args.steps % chunk or args.warmup % chunk:
        raise SystemExit("bench.py: --steps and --warmup must be multiples of --rollout-chunk (%d)" % chunk)
    lanes_per_agent = env.launch_info()["lanes_per_agent"]
    steps_per_launch = chunk if (chunk > 1 and env.launch_info()["rollout_one_launch"]) else 1   # (else ca_rollout is a loop of launches)

    def one_step(i):  # the production call: one ca_step (neighbours -> ORCA solve -> observation) ...
        if full:
            env._call("ca_step", env.h, pool[i % 16].data_ptr(), _lib.F_STATS | _lib.F_OBS)
        elif alan_mode:
            if i % chunk == 0:   # run_sim(mode=1) without the break: `chunk` online steps per call
                env._call("ca_alan_rollout", env.h, chunk, _lib.F_STATS)
        elif chunk == 1:
            env._call("ca_orca_step", env.h, _lib.F_STATS)
        elif i % chunk == 0:   # ... or one ca_rollout per `chunk` steps of an ORCA-only policy rollout
            env._call("ca_rollout", env.h, chunk, _lib.F_STATS)

    # warm-up: W steps, and on until MIN_WARM_SECONDS have passed (a 5-step warm-up leaves the clocks cold:
    # round 1's driver run read 84 us per kernel where a settled chip reads 78)
    tw = time.perf_counter()
    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    warm_run = args.warmup
    while time.perf_counter() - tw < MIN_WARM_SECONDS:
        for i in range(50 * chunk):
            one_step(warm_run + i)
        warm_run += 50 * chunk
        torch.cuda.synchronize()
    env.reset_stats()
    # The kernel launches of every 32nd step of the timed region (short runs: every (steps // 2)-th) carry a start and a stop event on
    # their own dispatch (hipExtLaunchKernel inside the library, on the stream the kernels run on).  A sampled step costs ~10 us of
    # dispatch serialisation (measured, round 5: sampling every step 119 us per step, every 8th 110.4, every 64th 109.3), so the
    # default run brackets 62 of its 2000 steps (0.3 % of the region) and the driver's --steps 20 two (0.9 %).
    # CA_BENCH_PROFILE_PERIOD overrides the period (diagnostic).
    env.profile(int(os.environ["CA_BENCH_PROFILE_PERIOD"]) if "CA_BENCH_PROFILE_PERIOD" in os.environ else
                (1 if steps_per_launch > 1 else max(1, min(32, args.steps // 2))))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(warm_run + i)
    torch.cuda.synchronize()
    own_ns = int((time.perf_counter() - t0) * 1e9)     # this rank's K steps alone (reported per rank; never the job's time)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ktimes = env.profile_read()
    env.profile(0)
    st = env.stats()
    steps_run = warm_run + args.steps      # every step this environment has been advanced since its scenario was drawn
    per_step_calls = None
    if not full and not alan_mode and chunk > 1:   # the same ORCA-only workload through one ca_orca_step call per step, for comparison
        torch.cuda.synchronize()
        n2 = max(chunk, min(args.steps, 500))
        t1 = time.perf_counter()
        for i in range(n2):
            env._call("ca_orca_step", env.h, _lib.F_STATS)
        torch.cuda.synchronize()
        per_step_calls = A * N * n2 / (time.perf_counter() - t1)
        steps_run += n2
    verified = None
    if args.verify > 0:
        # outside the timed region; a checker that cannot run (no oracle library, an exception inside it) must not discard a
        # measurement that completed: the line is printed with verified.bit_exact false + the error, and the exit code is non-zero
        try:
            if oracle_error is not None:
                raise RuntimeError("oracle library could not be built: " + oracle_error)
            verified = verify_against_oracle(env, args, w, p, scn, arena_offset, pool.cpu().numpy(), steps_run, steps_run - warm_run)
        except Exception as e:
            verified = {"arenas": 0, "first_global_arena": -1, "steps": steps_run, "timed_steps": steps_run - warm_run, "fields": 0,
                        "bit_exact": False, "mismatch": None, "error": "%s: %s" % (type(e).__name__, e)}

    # THE collective of the job: one all_gather of the per-rank record -- statistics, device, and the rank's own time for the
    # timed region in nanoseconds (RCCL over xGMI when N > 1); the job's time is the maximum over the gathered records
    per_rank_stats, total_stats = cad.gather_stats(st, device=coll_dev, extra={
        "device": local, "dt_ns": int(dt * 1e9), "verified": -1 if verified is None else int(verified["bit_exact"]),
        "verify_arena0": -1 if verified is None else verified["first_global_arena"], "warmup_steps_run": warm_run,
        # the rank's own time for its K steps BEFORE the closing barrier (dt_ns includes the wait for the slowest rank, so it
        # is nearly the same on every rank): this is the figure that shows which rank straggled
        "own_ns": own_ns})
    dt = max(d["dt_ns"] for d in per_rank_stats) * 1e-9
    if rank == 0:
        agents = A * N
        value = world * agents * args.steps / dt
        kbytes_of = {"nbr_kernel": 0, "step_kernel": BYTES_STEP_KERNEL_FULL if full else BYTES_STEP_KERNEL_ORCA,
                     "obs_kernel": BYTES_OBS_KERNEL}
        kms_of = {k: v[1] for k, v in ktimes.items() if v[0] > 0 and k in kbytes_of}   # (a T-step launch is reported per step)
        # dominant kernel: strictly the longest average launch (no tie-break)
        dom = max(kms_of, key=lambda k: kms_of[k])
        kms, kbytes = kms_of[dom], kbytes_of[dom]
        achieved = agents * kbytes / (kms * 1e-3) / 1e9
        per_kernel = {k: {"ms": round(v, 5), "algorithmic_bytes_per_agent": kbytes_of[k],
                          "algorithmic_GB_per_s": agents * kbytes_of[k] / (v * 1e-3) / 1e9,
                          "frac": agents * kbytes_of[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS}
                      for k, v in kms_of.items()}
        cj, csrc = counters_for(args.workload, args.mode, args.variant, args.starts, kms_of)
        traffic, traffic_low, traffic_high, traffic_detail, valu = None, None, None, {"source": csrc}, None
        if cj is not None:
            kk = cj.get("kernels", {})
            if dom in kk and "hbm_bytes_high" in kk[dom]:
                # FETCH_SIZE on gfx950 tallies the 128-B requests of wide reads at 64 B: the true figure lies between the
                # reported one (low) and the one with FETCH doubled (high); `traffic` quotes the conservative high end
                traffic_low, traffic_high = kk[dom].get("hbm_bytes_low"), kk[dom]["hbm_bytes_high"]
                traffic = traffic_high
            traffic_detail = {"source": csrc, "src_sha": cj.get("src_sha"),
                              "per_kernel": {k: {x: kk[k].get(x) for x in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes_low",
                                                                           "hbm_bytes_high", "algorithmic_bytes")}
                                             for k in kms_of if k in kk}}
            try:  # vector-ALU view: wave-level VALU instructions per launch x 64 lanes over the live kernel times
                cnt = {k: float(kk[k]["SQ_INSTS_VALU"]) for k in kms_of}
                lane_ops = 64.0 * sum(cnt.values())
                tsum = sum(kms_of.values()) * 1e-3

                def issue_cycles(c, cheap_other):
                    """Vector-issue cycles of one launch from the instruction-class counters and the measured issue
                    cost of each class on gfx950 (profiles/r02_valu_issue_rates_microbench.txt): v_add / v_mul_f32 2.7,
                    v_fma_f32 3.9, transcendental 8.3, conversions / 64-bit integer 4.2; 32-bit integer and the rest
                    (compares, selects, moves, min / max, DPP ...) between 2.7 (and, add, mov) and 4.2 (the others):
                    `cheap_other` picks the end of that range."""
                    known = ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                             "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT")
                    if not all(x in c for x in known):
                        return float(c["SQ_INSTS_VALU"]) * ISSUE_CYCLES_PER_VALU
                    other = float(c["SQ_INSTS_VALU"]) - sum(float(c[x]) for x in known)
                    flex = 2.7 if cheap_other else 4.2
                    return ((float(c["SQ_INSTS_VALU_ADD_F32"]) + float(c["SQ_INSTS_VALU_MUL_F32"])) * 2.7 +
                            float(c["SQ_INSTS_VALU_FMA_F32"]) * 3.9 + float(c["SQ_INSTS_VALU_TRANS_F32"]) * 8.3 +
                            (float(c["SQ_INSTS_VALU_INT64"]) + float(c["SQ_INSTS_VALU_CVT"])) * 4.2 +
                            (float(c["SQ_INSTS_VALU_INT32"]) + max(0.0, other)) * flex)
                bound_hi = {k: issue_cycles(kk[k], False) / N_SIMDS / CLOCK_HZ * 1e3 for k in kms_of}
                bound_lo = {k: issue_cycles(kk[k], True) / N_SIMDS / CLOCK_HZ * 1e3 for k in kms_of}
                valu = {"wave_insts_per_step": cnt, "lane_ops_per_s": lane_ops / tsum,
                        "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS, "frac": lane_ops / tsum / VALU_PEAK_LANE_OPS,
                        "per_kernel_frac": {k: 64.0 * cnt[k] / (kms_of[k] * 1e-3) / VALU_PEAK_LANE_OPS for k in kms_of},
                        # the bound that matters: every vector instruction of the launch issued back to back on its SIMD,
                        # priced per instruction class; [low, high] for the classes whose members differ in cost
                        "issue_bound_ms": bound_hi, "issue_bound_ms_low": bound_lo,
                        "frac_of_issue_bound": {k: bound_hi[k] / kms_of[k] for k in kms_of},
                        "frac_of_issue_bound_low": {k: bound_lo[k] / kms_of[k] for k in kms_of},
                        "source": "%s (SQ_INSTS_VALU and instruction-class counters per launch, same kernel sources and flags) "
                                  "over the live kernel times" % csrc}
            except Exception:
                valu = None
        limited_by = None
        if valu is not None and traffic is not None:
            hbm_frac = traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS
            limited_by = "valu-issue" if valu["frac_of_issue_bound_low"][dom] > hbm_frac else "hbm"
        out = {
            "metric": "agent-steps/sec (whole node), %d arenas x %d agents per GPU" % (A, N),
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "warmup_steps_run": warm_run, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("%s: %d arenas x %d agents per GPU, " + ("random start/goal crowd (ALAN recipe), " if scn.startswith("crowd") else "ALAN '" + scn + "' scenario, ") +
                                    "neighborDist %.1f, maxNeighbors %d, %s, %s, %s starts") %
                                   (args.workload, A, N, w["neighbor_dist"], w["max_neighbors"],
                                    "full env step (action -> ORCA -> reward/done -> laser obs)" if full
                                    else ("ALAN online step (softmax draw -> ORCA -> bandit update; no observation)" if alan_mode
                                          else "ORCA-only step (no observation)"),
                                    "boundary walls" if args.variant == "walls" else "obstacle-free", args.starts),
                       "mode": args.mode, "variant": args.variant, "starts": args.starts,
                       "rollout_chunk": chunk, "steps_per_launch": steps_per_launch, "lanes_per_agent": lanes_per_agent,
                       "sharding": "arenas, %d per GPU" % A},
            "world_size": world,
            "ranks": [{"rank": r, "device": d["device"], "agent_steps": d["agent_steps"], "verified": d["verified"],
                       "verify_arena0": d["verify_arena0"], "dt_ns": d["dt_ns"], "own_ns": d["own_ns"],
                       "warmup_steps_run": d["warmup_steps_run"],
                       "agent_steps_per_s": d["agent_steps"] / max(1, d["own_ns"]) * 1e9} for r, d in enumerate(per_rank_stats)],
            # slowest rank's own rate over the fastest's (1.0 = no straggler); from the per-rank times of the ONE all_gather
            "efficiency": (min(d["agent_steps"] / max(1, d["own_ns"]) for d in per_rank_stats) /
                           max(d["agent_steps"] / max(1, d["own_ns"]) for d in per_rank_stats)),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_low": traffic_low, "traffic_high": traffic_high,
                         # what the COUNTERS of this build say limits the dominant kernel: its vector-issue time (see `valu`)
                         # against the time its measured HBM traffic needs at the peak; null without a counter profile of
                         # these sources -- the HBM fraction above is reported because the metric asks for it
                         "limited_by": limited_by,
                         "algorithmic_bytes_per_agent": kbytes, "kernel_ms": kms, "per_kernel": per_kernel,
                         "traffic_detail": traffic_detail},
            "valu": valu,
            "kernels_ms": dict({k: round(v, 5) for k, v in kms_of.items()}, sum=round(sum(kms_of.values()), 5),
                               wall_per_step=dt / args.steps * 1e3,
                               sampled_launches={k: v[0] for k, v in ktimes.items() if v[0] > 0},
                               note="mean duration of the SAMPLED launches (start / stop events on their own dispatch: such a launch is "
                                    "serialised against its neighbours); unsampled launches overlap their ramp-up and drain with the "
                                    "neighbouring kernels by a few tenths of a microsecond, so wall_per_step may lie slightly below the sum"),
            "full_step_algorithmic": {"bytes_per_agent": 316 if full else 52,
                                      "GB_per_s": agents * (316 if full else 52) / (dt / args.steps) / 1e9,
                                      "frac": agents * (316 if full else 52) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
            "stats": {k: total_stats[k] for k in cad.STAT_KEYS},
            "launch": env.launch_info(),
            "src_sha": _b.loaded_sha(),      # compiled into the library that ran (ca_source_sha)
            "src_sha_on_disk": _b.source_sha(),
            "backend": backend if dist is not None else None,
            "collectives": {"data_path": 0, "job": "one all_gather of the per-rank record (statistics, device, dt)",
                            "timing_barriers": 2 if dist is not None else 0},
        }
        if verified is not None:      # rank 0's own record + the verdict of every rank (folded into the one all_gather)
            out["verified"] = dict(verified, bit_exact=all(d["verified"] == 1 for d in per_rank_stats),
                                   ranks_verified=sum(1 for d in per_rank_stats if d["verified"] == 1))
        if per_step_calls is not None:
            out["orca_per_step_calls"] = {"value": world * per_step_calls, "unit": "agent-steps/s",
                                          "note": "same workload, one ca_orca_step call (= one launch) per step; rank 0's rate x world"}
        # agent_steps is counted IN the solve kernels (per arena, the steps it was advanced): a launch that did not run, or
        # skipped arenas, shows here
        if sum(d["agent_steps"] for d in per_rank_stats) != world * agents * args.steps:
            raise SystemExit("bench.py: the ranks report %d agent-steps, expected %d" %
                             (sum(d["agent_steps"] for d in per_rank_stats), world * agents * args.steps))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.mode, args.cpu_seconds, args.variant, args.starts)
            try:   # the reference's OWN Python loop cannot run on this box (it never travels): quote the committed measurement
                j = json.load(open(os.path.join(ROOT, "profiles", "r05_reference_python_loop.json")))
                out["cpu_baseline"]["reference_python_loop"] = {
                    "label": j["label"], "measured": "in the build container by tools/time_reference_loop.py, NOT on this box",
                    "cpu": j["cpu"], "cores": j["cores_used"],
                    "agent_steps_per_s": {r["what"] + " [drawing " + r["drawing"] + "]": round(r["agent_steps_per_s"], 1) for r in j["rows"]}}
            except Exception:
                pass
        print(json.dumps(out))
        if verified is not None and not out["verified"]["bit_exact"]:
            raise SystemExit("bench.py: --verify FAILED: %s" % [verified.get("error") or verified["mismatch"]] + str([d["verified"] for d in per_rank_stats]))
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
