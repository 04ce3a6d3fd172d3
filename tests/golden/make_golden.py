#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own Python.

Runs only where /root/reference exists (the build container); the reference never travels and
none of its source is copied: the fixtures hold inputs and the outputs the reference computed.

  utils_vectors.npz   utils.py:5-40 line_intersection and utils.py:42-113 comp_laser on random
                      and edge-case inputs.
  env_doorway_*.npz   collision_avoidence_env.py Collision_Avoidance_Env driven through
                      __init__/reset/step/orca_step with the third-party `rvo2` module replaced
                      by oracle/rvo2_shim.py (the oracle's ORCA), `gym`, `ray`, `tkinter` stubbed,
                      time.clock/time.sleep shimmed and random.uniform replaced by a seeded
                      stream.  Pins the env loop A5-A9, A16-A19.
  alan_scenarios.npz  ALAN_true.py scenario generators: start/goal layouts and obstacle polygons.
  alan_online.npz     ALAN_true.py online_step runs (softmax selection, weights, arrival times, TTime).
  alan_orca.npz       ALAN_true.py run_sim(mode=0) episodes (orca_step + counter + done_test).
  alan_blocks.npz     three "blocks" worlds as the reference draws them (ALAN_true.py:333-374: four random blocks per
                      simulator, re-drawn by reset()), their polygons, and run_sim(mode=0) in each.

Usage: python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def load_ref_utils():
    spec = importlib.util.spec_from_file_location(
        "ref_utils", os.path.join(REF, "collision_avoidance/envs/utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def gen_utils_vectors():
    u = load_ref_utils()
    rng = np.random.RandomState(1234)
    # --- line_intersection: random + edge cases probed in SURVEY section 4 ---
    L1, L2 = [], []
    for _ in range(400):
        ray = ((0.0, 0.0), tuple(rng.uniform(-2, 2, 2)))
        seg = (tuple(rng.uniform(-2, 2, 2)), tuple(rng.uniform(-2, 2, 2)))
        L1.append(ray); L2.append(seg)
    for _ in range(100):   # general p0 too
        L1.append((tuple(rng.uniform(-1, 1, 2)), tuple(rng.uniform(-2, 2, 2))))
        L2.append((tuple(rng.uniform(-2, 2, 2)), tuple(rng.uniform(-2, 2, 2))))
    edge = [
        (((0, 0), (1.5, 0)), ((0, 1), (3, 1))),          # parallel
        (((0, 0), (1.5, 0)), ((1.5, -1), (1.5, 1))),     # hit exactly at the ray tip (t = 1)
        (((0, 0), (1.5, 0)), ((1.0, 0.0), (1.0, 1.0))),  # segment endpoint on the ray
        (((0, 0), (1.5, 0)), ((0.0, -1.0), (0.0, 1.0))), # segment through the origin -> d = 0
        (((0, 0), (1.5, 0)), ((-1.0, -1.0), (-1.0, 1.0))),  # behind
        (((0, 0), (1.5, 0)), ((1.0, 1.0), (1.0, -1.0))), # reversed orientation (denom < 0)
        (((0, 0), (1.5, 0)), ((0.5, 0.0), (1.0, 0.0))),  # collinear overlapping
        (((0, 0), (1.5, 0)), ((1.6, -1), (1.6, 1))),     # just beyond the tip
    ]
    for a, b in edge:
        L1.append(a); L2.append(b)
    li_in = np.array([[a[0][0], a[0][1], a[1][0], a[1][1], b[0][0], b[0][1], b[1][0], b[1][1]]
                      for a, b in zip(L1, L2)], np.float64)
    li_out = []
    for a, b in zip(L1, L2):
        d, p = u.line_intersection(a, b)
        li_out.append([d, p[0], p[1]])
    li_out = np.array(li_out, np.float64)

    # --- comp_laser: the reference env's own ray table and octagon are rebuilt here from the
    # formulas' constants (nd = 1.5, 16 rays; r = 0.5, 8 chords) with the reference's math calls
    from math import cos, sin, pi
    nd, nr = 1.5, 16
    rays = [((0, 0), (nd * cos(i * 2 * pi / nr), -nd * sin(i * 2 * pi / nr))) for i in range(nr)]
    cases_segs, cases_orient, cases_out = [], [], []

    def octagon(rel, r=0.5, n=8):
        pts = [(r * cos(i * 2 * pi / n), -r * sin(i * 2 * pi / n)) for i in range(n)]
        return [((pts[i][0] + rel[0], pts[i][1] + rel[1]),
                 (pts[(i + 1) % n][0] + rel[0], pts[(i + 1) % n][1] + rel[1])) for i in range(n)]

    def run_case(lines_with_vel, orient):
        res = u.comp_laser(rays, lines_with_vel, orient)
        out = np.array([[r[0][0], r[0][1], r[1][0], r[1][1]] for r in res], np.float64)
        segs = np.array([[l[0][0][0], l[0][0][1], l[0][1][0], l[0][1][1], l[1][0], l[1][1]]
                         for l in lines_with_vel], np.float64)
        cases_segs.append(segs); cases_orient.append(orient); cases_out.append(out)

    # the worked example of SURVEY section 8c
    lw = [(s, (0.2, -0.4)) for s in octagon((1.0, 0.3))] + [(((-0.7, -3.0), (-0.7, 3.0)), (0, 0))]
    run_case(lw, (0.6, 0.8))
    for _ in range(60):
        lw = []
        for _k in range(rng.randint(1, 6)):
            ang, dist = rng.uniform(0, 2 * pi), rng.uniform(0.2, 1.9)
            vel = tuple(rng.uniform(-1, 1, 2))
            lw += [(s, vel) for s in octagon((dist * cos(ang), dist * sin(ang)))]
        for _k in range(rng.randint(0, 3)):
            p = rng.uniform(-2, 2, 4)
            lw.append((((p[0], p[1]), (p[2], p[3])), (0, 0)))
        th = rng.uniform(0, 2 * pi)
        run_case(lw, (cos(th), sin(th)))
    # segment through the origin: d = 0, hit (0,0) -> reported as a miss (utils.py:103)
    run_case([(((0.0, -1.0), (0.0, 1.0)), (0.3, 0.3)), (((1.0, -1.0), (1.0, 1.0)), (0.1, 0.2))], (1.0, 0.0))
    m = max(s.shape[0] for s in cases_segs)
    segs = np.zeros((len(cases_segs), m, 6)); counts = np.zeros(len(cases_segs), np.int32)
    for i, s in enumerate(cases_segs):
        segs[i, :s.shape[0]] = s; counts[i] = s.shape[0]
    np.savez_compressed(os.path.join(HERE, "utils_vectors.npz"),
                        li_in=li_in, li_out=li_out,
                        rays=np.array([r[1] for r in rays], np.float64),
                        cl_segs=segs, cl_counts=counts,
                        cl_orient=np.array(cases_orient, np.float64),
                        cl_out=np.array(cases_out, np.float64))
    print("utils_vectors.npz: %d line_intersection, %d comp_laser cases" % (len(li_in), len(cases_out)))


# ---------------------------------------------------------------------------------------------
class _Stream:
    """Seeded replacement for random.uniform; records what it hands out."""

    def __init__(self, seed, box=None):
        self.rng = np.random.RandomState(seed)
        self.box = box

    def __call__(self, a, b):
        return float(a + (b - a) * self.rng.random_sample())


def install_stubs():
    from oracle import rvo2_shim
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class Box(object):
        def __init__(self, low, high, shape):
            self.low, self.high, self.shape = low, high, shape

    gym.Env = Env
    gym.spaces = types.ModuleType("gym.spaces"); gym.spaces.Box = Box
    gym.utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    gym.utils.seeding = seeding
    gym.envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration"); reg.register = lambda **kw: None
    gym.envs.registration = reg
    for name, mod in [("gym", gym), ("gym.spaces", gym.spaces), ("gym.utils", gym.utils),
                      ("gym.utils.seeding", seeding), ("gym.envs", gym.envs),
                      ("gym.envs.registration", reg)]:
        sys.modules[name] = mod
    ray = types.ModuleType("ray")
    for n in ("ray.rllib", "ray.rllib.env", "ray.rllib.env.multi_agent_env"):
        sys.modules[n] = types.ModuleType(n)
    sys.modules["ray"] = ray

    class MultiAgentEnv(object):
        pass

    sys.modules["ray.rllib.env.multi_agent_env"].MultiAgentEnv = MultiAgentEnv
    tk = types.ModuleType("tkinter")

    class _Widget(object):
        def __init__(self, *a, **k):
            self._n = 0

        def __getattr__(self, name):
            def f(*a, **k):
                self._n += 1
                return self._n
            return f

    tk.Tk = _Widget; tk.Canvas = _Widget; tk.LAST = "last"
    tk.__all__ = ["Tk", "Canvas", "LAST"]
    sys.modules["tkinter"] = tk
    rv = types.ModuleType("rvo2"); rv.PyRVOSimulator = rvo2_shim.PyRVOSimulator
    sys.modules["rvo2"] = rv
    if not hasattr(time, "clock"):
        time.clock = time.perf_counter
    time.sleep = lambda s: None
    sys.path.insert(0, REF)


def sim_state(env):
    n = env.numAgents
    pos = np.array([env.sim.getAgentPosition(i) for i in range(n)], np.float32)
    vel = np.array([env.sim.getAgentVelocity(i) for i in range(n)], np.float32)
    pref = np.array([env.sim.getAgentPrefVelocity(i) for i in range(n)], np.float32)
    tgt = np.array(env.world["targets_pos"], np.float64)
    return pos, vel, pref, tgt


def obs_array(env, d):
    return np.array([d['agent_%d' % i] for i in range(env.numAgents)], np.float64)


def gen_env_fixture(name, n_agents, seed, n_steps, reset_at, obs_every, spawn_squeeze=None, act_scale=None,
                    orca_every=17, reset_on_done=False, after_reset_steps=0):
    """act_scale: actions ~ U(-act_scale, act_scale) (agents pass the door); reset_on_done: the caller's protocol
    (run_rllib.py's workers): the step after '__all__' comes back True is preceded by env.reset(), and the run ends
    after_reset_steps steps later -- the END of an episode: env.py:352-365, 404-414, 461-488."""
    import warnings
    warnings.simplefilter("ignore")
    import collision_avoidance.envs.collision_avoidence_env as refenv
    stream = _Stream(seed)
    if spawn_squeeze is not None:
        # still a legal outcome of uniform(a, b): draws are squeezed towards the doorway so that
        # walls, the door gap and neighbours are all exercised within a short run
        base = stream

        def squeezed(a, b):
            u = base.rng.random_sample()
            lo, hi = spawn_squeeze.get((a, b), (0.0, 1.0))
            return float(a + (b - a) * (lo + (hi - lo) * u))
        refenv.uniform = squeezed
    else:
        refenv.uniform = stream
    import io
    import contextlib
    env = refenv.Collision_Avoidance_Env(numAgents=n_agents)
    rng = np.random.RandomState(seed + 1)
    rec = dict(actions=[], pos=[], vel=[], pref=[], tgt=[], reward=[], done_all=[], agents_done=[],
               obs=[], obs_steps=[], reset_steps=[], reset_pos=[], reset_obs=[], kind=[])
    p, v, pf, t = sim_state(env)
    init = dict(pos0=p, vel0=v, pref0=pf, tgt0=t, obs0=obs_array(env, env.gym_obs))
    pending_reset, last = False, n_steps
    rec["step_count"] = []
    for s in range(n_steps):
        if s >= last:
            break
        if s in reset_at or pending_reset:
            pending_reset = False
            with contextlib.redirect_stdout(io.StringIO()):
                o = env.reset()
            p, v, pf, t = sim_state(env)
            rec["reset_steps"].append(s); rec["reset_pos"].append(p); rec["reset_obs"].append(obs_array(env, o))
        kind = 0
        if orca_every and s % orca_every == orca_every - 1:
            kind = 1  # an orca_step (env.py:447-458) in between
            with contextlib.redirect_stdout(io.StringIO()):
                env.orca_step((0, 0))
            act = np.zeros(n_agents, np.float32)
            rew = np.zeros(n_agents)
            done_all = False
            o = env.gym_obs
        else:
            if act_scale is None:
                act = rng.uniform(-np.pi, np.pi, n_agents).astype(np.float32) * (0.35 if s % 3 else 1.0)
            else:
                act = rng.uniform(-act_scale, act_scale, n_agents).astype(np.float32)
            with contextlib.redirect_stdout(io.StringIO()):
                o, r, d, _ = env.step({'agent_%d' % i: act[i:i + 1] for i in range(n_agents)})
            rew = np.array([float(r['agent_%d' % i]) for i in range(n_agents)], np.float64)
            done_all = bool(d['__all__'])
            if done_all and reset_on_done and last == n_steps:
                pending_reset, last = True, s + 1 + after_reset_steps
        rec["step_count"].append(env.step_count)
        p, v, pf, t = sim_state(env)
        rec["kind"].append(kind); rec["actions"].append(act); rec["pos"].append(p); rec["vel"].append(v)
        rec["pref"].append(pf); rec["tgt"].append(t); rec["reward"].append(rew)
        rec["done_all"].append(done_all); rec["agents_done"].append(np.array(env.agents_done, np.int32))
        if s % obs_every == 0 or s < 24 or kind == 1:
            rec["obs"].append(obs_array(env, o)); rec["obs_steps"].append(s)
    if not reset_on_done:
        del rec["step_count"]         # the fixtures of rounds 1-4 regenerate byte for byte
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(init)
    out["n_agents"] = np.int32(n_agents)
    out["step_count_final"] = np.int32(env.step_count)
    np.savez_compressed(os.path.join(HERE, name), **out)
    nz = int((np.abs(out["obs"]) > 0).sum())
    print("%s: %d steps, %d obs snapshots (%d non-zero obs entries), done_all any=%s, agents done %d, resets at %s"
          % (name, len(rec["kind"]), len(rec["obs_steps"]), nz, bool(np.any(out["done_all"])),
             int(out["agents_done"].max(axis=0).sum()), rec["reset_steps"]))


def gen_alan_scenarios():
    import warnings
    warnings.simplefilter("ignore")
    import collision_avoidance.ALAN.ALAN_true as alan
    out = {}
    cases = (("circle", 8), ("circle", 100), ("crowd", 16), ("congested", 24), ("incoming", 17),
             ("incoming", 26), ("blocks", 12), ("deadlock", 20), ("deadlock", 9))
    for scen, n in cases:
        alan.uniform = _Stream(77)
        sim = alan.Collision_Avoidance_Sim(numAgents=n, scenario=scen, visualize=False)
        key = "%s%d_" % (scen, n)
        out[key + "pos"] = np.array([sim.sim.getAgentPosition(i) for i in range(n)], np.float32)
        out[key + "goal"] = np.array([sim.world["targets_pos"][i][0] for i in range(n)], np.float64)
        out[key + "goal2"] = np.array([sim.world["targets_pos"][i][1] for i in range(n)], np.float64)
        out[key + "envsize"] = np.float64(sim.envsize)
        out[key + "max_step"] = np.int32(sim.max_step)
        out[key + "obst"] = np.array([[sim.sim.getObstacleVertex(v) for v in ids]
                                      for ids in sim.world["obstacles_vertex_ids"]], np.float32)
    np.savez_compressed(os.path.join(HERE, "alan_scenarios.npz"), **out)
    print("alan_scenarios.npz:", len(out), "arrays for", [c[0] + str(c[1]) for c in cases])


def gen_alan_online():
    """ALAN_true.py online_step (softmax action selection + bandit update, :569-628) and the run_sim
    bookkeeping (:119-131) driven for a few hundred steps; np.random.choice is replaced by a
    recording re-implementation of numpy's own draw (cdf = cumsum(p); cdf /= cdf[-1];
    searchsorted(u, side='right')) fed from a seeded uniform stream."""
    import warnings
    warnings.simplefilter("ignore")
    import collision_avoidance.ALAN.ALAN_true as alan
    real_choice = np.random.choice
    out = {}
    cases = (("crowd", 12, None, 360), ("circle", 8, [(1, 0), (-0.946001067452245, -0.3241635087100537),
                                                      (0.9651519613079926, -0.2616900677964968)], 300),
             ("deadlock", 10, [(1, 0), (-0.8507885983503763, -0.5255080978605392)], 260),
             ("crowd", 9, [(1, 0), (0.06130798855686512, -0.9981188959934139), (0.6957331792402002, -0.718300315539624),
                           (0.2199481143650689, 0.9755115719391804), (-0.34342388205007957, -0.9391805136594631),
                           (-0.5032837686361407, -0.8641211999641044), (0.23892114215448593, 0.9710389733844857),
                           (-0.9579667593692613, -0.28687922187491344), (-0.6767380373137093, 0.7362238985884584)], 200))
    for ci, (scen, n, actions, steps) in enumerate(cases):
        alan.uniform = _Stream(500 + ci)
        urng = np.random.RandomState(900 + ci)
        drawn = []

        def choice(a, size=None, p=None):
            cdf = np.asarray(p, np.float64).cumsum()
            cdf /= cdf[-1]
            u = urng.random_sample()
            drawn.append(u)
            return np.array([cdf.searchsorted(u, side='right')])
        np.random.choice = choice
        try:
            sim = alan.Collision_Avoidance_Sim(numAgents=n, scenario=scen, online_actions=actions, visualize=False)
            sim.reset(actions)   # the constructor zeroes min_TTime after computing it (:71); reset() keeps it
            key = "c%d_" % ci
            nA = len(sim.online_actions)
            out[key + "scenario"] = np.array(scen)
            out[key + "actions"] = np.array(sim.online_actions, np.float64)
            out[key + "pos0"] = np.array([sim.sim.getAgentPosition(i) for i in range(n)], np.float32)
            out[key + "vel0"] = np.array([sim.sim.getAgentVelocity(i) for i in range(n)], np.float32)
            out[key + "goal0"] = np.array([sim.world["targets_pos"][i][0] for i in range(n)], np.float64)
            out[key + "goal20"] = np.array([sim.world["targets_pos"][i][1] for i in range(n)], np.float64)
            rec = dict(u=[], pos=[], vel=[], w=[], t=[], done=[], times=[])
            for s_ in range(steps):
                del drawn[:]
                sim.online_step()
                sim.step_count += 1
                sim.done_test()
                rec["u"].append(list(drawn))
                rec["pos"].append([sim.sim.getAgentPosition(i) for i in range(n)])
                rec["vel"].append([sim.sim.getAgentVelocity(i) for i in range(n)])
                rec["w"].append(np.array(sim.world["action_weights"], np.float64))
                rec["t"].append(np.array(sim.world["action_times"], np.float64))
                rec["done"].append(list(sim.agents_done))
                rec["times"].append(list(sim.agents_time))
            out[key + "u"] = np.array(rec["u"], np.float64)
            out[key + "pos"] = np.array(rec["pos"], np.float32)
            out[key + "vel"] = np.array(rec["vel"], np.float32)
            out[key + "w"] = np.array(rec["w"], np.float64)[::10]          # every 10th step
            out[key + "t"] = np.array(rec["t"], np.float64)[::10]
            out[key + "w_last"] = np.array(rec["w"][-1], np.float64)
            out[key + "done"] = np.array(rec["done"], np.int32)
            out[key + "agents_time"] = np.array(rec["times"][-1], np.float64)
            times = np.array(sim.agents_time)
            out[key + "TTime"] = np.float64(np.average(times) + 3 * np.std(times, 0))   # ALAN:127-130
            out[key + "min_TTime"] = np.float64(sim.min_TTime)
            out[key + "max_step"] = np.int32(sim.max_step)
        finally:
            np.random.choice = real_choice
    out["n_cases"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(HERE, "alan_online.npz"), **out)
    print("alan_online.npz: %d cases, done counts %s" % (len(cases), [int(out["c%d_done" % i][-1].sum()) for i in range(len(cases))]))


def gen_alan_orca():
    """ALAN_true.py run_sim(mode=0): orca_step (:631-636) + step counter + done_test (:118-121) until every agent
    has arrived, i.e. the reference's own plain-ORCA episode loop, with its return values."""
    import warnings
    warnings.simplefilter("ignore")
    import collision_avoidance.ALAN.ALAN_true as alan
    out = {}
    cases = (("crowd", 10), ("circle", 12), ("incoming", 10), ("congested", 8))
    for ci, (scen, n) in enumerate(cases):
        alan.uniform = _Stream(700 + ci)
        sim = alan.Collision_Avoidance_Sim(numAgents=n, scenario=scen, visualize=False)
        sim.reset(None)
        key = "c%d_" % ci
        out[key + "scenario"] = np.array(scen)
        out[key + "pos0"] = np.array([sim.sim.getAgentPosition(i) for i in range(n)], np.float32)
        out[key + "vel0"] = np.array([sim.sim.getAgentVelocity(i) for i in range(n)], np.float32)
        out[key + "goal0"] = np.array([sim.world["targets_pos"][i][0] for i in range(n)], np.float64)
        out[key + "goal20"] = np.array([sim.world["targets_pos"][i][1] for i in range(n)], np.float64)
        out[key + "pref0"] = np.array([sim.sim.getAgentPrefVelocity(i) for i in range(n)], np.float32)
        rec = dict(pos=[], vel=[], done=[])
        steps = 0
        cap = min(sim.max_step, 1500)
        for s_ in range(cap):                       # the body of run_sim(0), recorded step by step
            sim.orca_step()
            sim.step_count += 1
            success = sim.done_test()
            steps += 1
            rec["pos"].append([sim.sim.getAgentPosition(i) for i in range(n)])
            rec["vel"].append([sim.sim.getAgentVelocity(i) for i in range(n)])
            rec["done"].append(list(sim.agents_done))
            if success:
                break
        out[key + "pos"] = np.array(rec["pos"], np.float32)[::5]       # every 5th step
        out[key + "vel"] = np.array(rec["vel"], np.float32)[::5]
        out[key + "pos_last"] = np.array(rec["pos"][-1], np.float32)
        out[key + "done"] = np.array(rec["done"], np.int32)[::5]
        out[key + "done_last"] = np.array(rec["done"][-1], np.int32)
        out[key + "steps"] = np.int32(steps)
        out[key + "success"] = np.int32(bool(success))
        out[key + "agents_time"] = np.array(sim.agents_time, np.float64)
        times = np.array(sim.agents_time)
        out[key + "TTime"] = np.float64(np.average(times) + 3 * np.std(times, 0))
        out[key + "max_step"] = np.int32(sim.max_step)
    out["n_cases"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(HERE, "alan_orca.npz"), **out)
    print("alan_orca.npz: %d cases, steps %s, success %s" % (len(cases), [int(out["c%d_steps" % i]) for i in range(len(cases))],
                                                             [int(out["c%d_success" % i]) for i in range(len(cases))]))


def gen_alan_blocks():
    """Three random "blocks" worlds (ALAN_true.py:359-372 draws four blocks per simulator; reset() -- :92-100 -- draws
    new ones) and the plain-ORCA episode of each: what a batch with one world per arena has to reproduce."""
    import warnings
    warnings.simplefilter("ignore")
    import collision_avoidance.ALAN.ALAN_true as alan
    out = {}
    n = 12
    alan.uniform = _Stream(4242)
    sim = alan.Collision_Avoidance_Sim(numAgents=n, scenario="blocks", visualize=False)
    for wi in range(3):
        sim.reset(None)                                 # a new world from the same stream, like the trainer's rounds
        key = "w%d_" % wi
        out[key + "obst"] = np.array([[sim.sim.getObstacleVertex(v) for v in ids]
                                      for ids in sim.world["obstacles_vertex_ids"]], np.float32)   # [5, 4, 2]
        out[key + "pos0"] = np.array([sim.sim.getAgentPosition(i) for i in range(n)], np.float32)
        out[key + "vel0"] = np.array([sim.sim.getAgentVelocity(i) for i in range(n)], np.float32)
        out[key + "goal0"] = np.array([sim.world["targets_pos"][i][0] for i in range(n)], np.float64)
        out[key + "goal20"] = np.array([sim.world["targets_pos"][i][1] for i in range(n)], np.float64)
        out[key + "pref0"] = np.array([sim.sim.getAgentPrefVelocity(i) for i in range(n)], np.float32)
        rec = dict(pos=[], vel=[], done=[])
        steps, success = 0, False
        for s_ in range(min(sim.max_step, 1500)):      # the body of run_sim(0), recorded step by step
            sim.orca_step()
            sim.step_count += 1
            success = sim.done_test()
            steps += 1
            rec["pos"].append([sim.sim.getAgentPosition(i) for i in range(n)])
            rec["vel"].append([sim.sim.getAgentVelocity(i) for i in range(n)])
            rec["done"].append(list(sim.agents_done))
            if success:
                break
        out[key + "pos"] = np.array(rec["pos"], np.float32)[::5]
        out[key + "vel"] = np.array(rec["vel"], np.float32)[::5]
        out[key + "pos_last"] = np.array(rec["pos"][-1], np.float32)
        out[key + "done"] = np.array(rec["done"], np.int32)[::5]
        out[key + "done_last"] = np.array(rec["done"][-1], np.int32)
        out[key + "steps"] = np.int32(steps)
        out[key + "success"] = np.int32(bool(success))
        out[key + "agents_time"] = np.array(sim.agents_time, np.float64)
        out[key + "max_step"] = np.int32(sim.max_step)
        out[key + "n_obst_vertices"] = np.int32(sim.sim.getNumObstacleVertices()) if hasattr(sim.sim, "getNumObstacleVertices") else np.int32(-1)
    out["n_worlds"] = np.int32(3)
    out["n_agents"] = np.int32(n)
    np.savez_compressed(os.path.join(HERE, "alan_blocks.npz"), **out)
    print("alan_blocks.npz: steps %s, success %s, block 0 of each world at %s" % (
        [int(out["w%d_steps" % i]) for i in range(3)], [int(out["w%d_success" % i]) for i in range(3)],
        [out["w%d_obst" % i][1][0].tolist() for i in range(3)]))

def gen_alan_finished():
    """Episodes that END, through the reference's OWN loop: run_sim(mode) (ALAN_true.py:106-131) is called as it is --
    the recorder sits in done_test, which the loop calls once per step after its counter -- for runs where every agent
    arrives (success -> break, :121-123; total_time, TTime over a finished run, :125-131) and one that runs into
    max_step with an agent still on its way.  mode 1 = online_step with numpy's choice replaced as in gen_alan_online."""
    import warnings
    warnings.simplefilter("ignore")
    import collision_avoidance.ALAN.ALAN_true as alan
    real_choice = np.random.choice
    out = {}
    acts3 = [(1, 0), (0.7071067811865476, 0.7071067811865476), (0.7071067811865476, -0.7071067811865476)]
    cases = (("circle", 8, 0, None), ("crowd", 10, 0, None), ("incoming", 5, 0, None), ("congested", 6, 0, None),
             ("crowd", 6, 0, None), ("circle", 8, 1, acts3), ("crowd", 10, 1, None))
    seeds = (811, 814, 812, 813, 810, 821, 824)
    for ci, ((scen, n, mode, actions), seed) in enumerate(zip(cases, seeds)):
        alan.uniform = _Stream(seed)
        urng = np.random.RandomState(seed + 100)
        drawn = []

        def choice(a, size=None, p=None):
            cdf = np.asarray(p, np.float64).cumsum()
            cdf /= cdf[-1]
            u = urng.random_sample()
            drawn.append(u)
            return np.array([cdf.searchsorted(u, side='right')])
        np.random.choice = choice
        try:
            sim = alan.Collision_Avoidance_Sim(numAgents=n, scenario=scen, online_actions=actions, visualize=False)
            sim.reset(actions)
            key = "c%d_" % ci
            out[key + "scenario"] = np.array(scen)
            out[key + "mode"] = np.int32(mode)
            out[key + "actions"] = np.array(sim.online_actions, np.float64)
            out[key + "pos0"] = np.array([sim.sim.getAgentPosition(i) for i in range(n)], np.float32)
            out[key + "vel0"] = np.array([sim.sim.getAgentVelocity(i) for i in range(n)], np.float32)
            out[key + "pref0"] = np.array([sim.sim.getAgentPrefVelocity(i) for i in range(n)], np.float32)
            out[key + "goal0"] = np.array([sim.world["targets_pos"][i][0] for i in range(n)], np.float64)
            out[key + "goal20"] = np.array([sim.world["targets_pos"][i][1] for i in range(n)], np.float64)
            rec = dict(u=[], pos=[], vel=[], done=[])
            inner = sim.done_test

            def recording_done_test():
                r = inner()
                rec["u"].append(list(drawn)); del drawn[:]
                rec["pos"].append([sim.sim.getAgentPosition(i) for i in range(n)])
                rec["vel"].append([sim.sim.getAgentVelocity(i) for i in range(n)])
                rec["done"].append(list(sim.agents_done))
                return r
            sim.done_test = recording_done_test
            success, total_time, ttime, min_ttime = sim.run_sim(mode)      # the reference's loop itself
            every = 5
            out[key + "u"] = np.array(rec["u"], np.float64)                  # [steps, n] (mode 1) or [steps, 0]
            out[key + "pos"] = np.array(rec["pos"], np.float32)[::every]
            out[key + "vel"] = np.array(rec["vel"], np.float32)[::every]
            out[key + "done"] = np.array(rec["done"], np.int32)
            out[key + "pos_last"] = np.array(rec["pos"][-1], np.float32)
            out[key + "vel_last"] = np.array(rec["vel"][-1], np.float32)
            out[key + "steps"] = np.int32(sim.step_count)
            out[key + "success"] = np.int32(bool(success))
            out[key + "total_time"] = np.float64(total_time)
            out[key + "TTime"] = np.float64(ttime)
            out[key + "min_TTime"] = np.float64(min_ttime)
            out[key + "agents_time"] = np.array(sim.agents_time, np.float64)
            out[key + "goal_last"] = np.array([sim.world["targets_pos"][i][0] for i in range(n)], np.float64)
            out[key + "max_step"] = np.int32(sim.max_step)
            if mode == 1:
                out[key + "w_last"] = np.array(sim.world["action_weights"], np.float64)
                out[key + "t_last"] = np.array(sim.world["action_times"], np.float64)
        finally:
            np.random.choice = real_choice
    out["n_cases"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(HERE, "alan_finished.npz"), **out)
    print("alan_finished.npz: %d cases, steps %s, success %s" % (
        len(cases), [int(out["c%d_steps" % i]) for i in range(len(cases))], [int(out["c%d_success" % i]) for i in range(len(cases))]))


def gen_round5():
    """The END of an episode (round 5): agents_done / the (-10, 5) retarget / '__all__' by the step cap and by the
    last arrival / reset() after an episode keeping the swapped targets; run_sim episodes that finish."""
    gen_env_fixture("env_doorway_n6_episode.npz", 6, seed=21, n_steps=1200, reset_at=(), obs_every=10, act_scale=0.15,
                    orca_every=None, reset_on_done=True, after_reset_steps=100)
    gen_env_fixture("env_doorway_n4_all_done.npz", 4, seed=24, n_steps=1200, reset_at=(), obs_every=10, act_scale=0.05,
                    orca_every=None, reset_on_done=True, after_reset_steps=100)
    gen_alan_finished()


if __name__ == "__main__":
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "blocks":   # only the file added in round 2
        install_stubs()
        gen_alan_blocks()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round5":   # only the files added in round 5
        install_stubs()
        gen_round5()
        sys.exit(0)
    gen_utils_vectors()
    install_stubs()
    gen_env_fixture("env_doorway_n10.npz", 10, seed=7, n_steps=420, reset_at=(200,), obs_every=6)
    gen_env_fixture("env_doorway_n6_dense.npz", 6, seed=11, n_steps=260, reset_at=(130,), obs_every=4,
                    spawn_squeeze={(5.0, 10): (0.0, 0.25), (0, 10): (0.3, 0.7)})
    gen_alan_scenarios()
    gen_alan_online()
    gen_alan_orca()
    gen_alan_blocks()
    gen_round5()
