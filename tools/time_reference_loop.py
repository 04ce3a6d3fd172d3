#!/usr/bin/env python3
"""Times the REFERENCE's own Python loop in this container -- the CPU path BASELINE config C1 names.

What runs is /root/reference's code itself: Collision_Avoidance_Env.step() / orca_step() (collision_avoidence_env.py:367-416,
447-458, the __main__ loop :570-573) with utils.py's comp_laser / line_intersection underneath (utils.py:5-113), and
ALAN_true.Collision_Avoidance_Sim(numAgents=8, scenario="circle").run_sim(0 | 1) (ALAN_true.py:106-131).  What does NOT
run is the third-party `rvo2` module (absent from the image, un-vendored: SURVEY section 8c): oracle/rvo2_shim.py stands
in for it -- the oracle's ORCA behind per-scalar ctypes calls, which are slower per call than Cython's, so the figure is
"reference Python over the oracle's ORCA", not Python-RVO2.  gym / ray / tkinter are stubbed exactly as for the golden
fixtures (tests/golden/make_golden.py install_stubs); drawing is measured both ways: FLAG_DRAW off (and orca_step's
unconditional draw_update made a no-op) and on (against the no-op Tk stub; the shipped env also SLEEPS 1/60 s per step,
collision_avoidence_env.py:567 -- the stub's sleep returns at once, so the shipped figure is min(this, 60) steps/s).

Nothing of the reference travels; where /root/reference is absent (the GPU box) this prints a note and exits 0.
  python tools/time_reference_loop.py [out.json] [seconds per cell]
"""
import contextlib
import io
import json
import os
import platform
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference"


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor()


def timed(fn, seconds, chunk):
    """Calls fn() in chunks of `chunk` until `seconds` have passed; returns (calls, elapsed)."""
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(chunk):
            fn()
        n += chunk
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return n, dt


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_reference_python_loop.json")
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    if not os.path.isdir(REF):
        print("reference not present (GPU box?): nothing to time")
        return 0
    warnings.simplefilter("ignore")
    import numpy as np
    import make_golden as mg
    mg.install_stubs()
    import collision_avoidance.envs.collision_avoidence_env as refenv
    import collision_avoidance.ALAN.ALAN_true as alan
    rows = []
    quiet = contextlib.redirect_stdout(io.StringIO())

    def env_rows(draw):
        refenv.FLAG_DRAW = draw
        refenv.uniform = mg._Stream(7)
        with quiet:
            env = refenv.Collision_Avoidance_Env()            # the default: 10 agents, the doorway world (env.py:26, 77-123)
        if not draw:
            env.draw_update = lambda: None                    # orca_step draws unconditionally (env.py:458)
        n = env.numAgents
        rng = np.random.RandomState(8)
        acts = [{'agent_%d' % i: rng.uniform(-0.3, 0.3, 1).astype(np.float32) for i in range(n)} for _ in range(64)]
        k = [0]

        def step():
            with quiet:
                _, _, d, _ = env.step(acts[k[0] % 64])
                k[0] += 1
                if d['__all__']:
                    env.reset()

        def orca():
            with quiet:
                env.orca_step((0, 0))
        for name, fn, cite in (("env.step(action_dict)", step, "collision_avoidence_env.py:367-416"),
                               ("env.orca_step() -- the __main__ loop", orca, "collision_avoidence_env.py:447-458, 570-573")):
            timed(fn, 0.5, 10)
            calls, dt = timed(fn, seconds, 20)
            rows.append({"what": name, "cite": cite, "agents": n, "world": "doorway", "drawing": "Tk stub (no sleep)" if draw else "off",
                         "steps": calls, "seconds": round(dt, 3), "env_steps_per_s": calls / dt, "agent_steps_per_s": calls * n / dt})
    env_rows(False)
    env_rows(True)

    # where the time goes (the reference's dominant cost): line_intersection calls per step
    import cProfile
    import pstats
    refenv.FLAG_DRAW = False
    refenv.uniform = mg._Stream(7)
    with quiet:
        env = refenv.Collision_Avoidance_Env()
    rng = np.random.RandomState(8)
    pr = cProfile.Profile()
    pr.enable()
    with quiet:
        for s in range(100):
            env.step({'agent_%d' % i: rng.uniform(-0.3, 0.3, 1).astype(np.float32) for i in range(env.numAgents)})
    pr.disable()
    st = pstats.Stats(pr)
    prof = {}
    for (f, ln, fn), (cc, nc, tt, ct, _) in st.stats.items():
        if fn in ("line_intersection", "comp_laser", "_get_obs", "step", "doStep"):
            prof[fn] = {"calls_per_100_steps": nc, "cumulative_s": round(ct, 3)}

    for mode, label in ((0, "run_sim(0): plain ORCA"), (1, "run_sim(1): ALAN online learning")):
        n_runs, steps, agents, t_all = 0, 0, 8, 0.0
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            alan.uniform = mg._Stream(30 + n_runs)
            np.random.seed(40 + n_runs)
            sim = alan.Collision_Avoidance_Sim(numAgents=agents, scenario="circle", visualize=False)   # BASELINE config C1's words
            sim.reset(None)
            t0 = time.perf_counter()
            sim.run_sim(mode)
            t_all += time.perf_counter() - t0
            steps += sim.step_count
            n_runs += 1
        rows.append({"what": "ALAN_true.Collision_Avoidance_Sim(8, 'circle')." + label, "cite": "ALAN_true.py:106-131, 569-636",
                     "agents": agents, "world": "circle", "drawing": "off (visualize=False)", "steps": steps, "episodes": n_runs,
                     "seconds": round(t_all, 3), "env_steps_per_s": steps / t_all, "agent_steps_per_s": steps * agents / t_all})
    rec = {"label": "reference Python over the oracle's ORCA (rvo2 absent), build container, one core",
           "what_runs": "the reference's own source files, imported from /root/reference; rvo2 -> oracle/rvo2_shim.py (ctypes); "
                        "gym, ray, tkinter stubbed (tests/golden/make_golden.py install_stubs)",
           "nproc": os.cpu_count(), "cores_used": 1, "cpu": cpu_model(), "python": platform.python_version(),
           "numpy": np.__version__, "seconds_per_cell": seconds, "rows": rows, "cprofile_100_steps_of_env_step": prof}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    for r in rows:
        print("%-72s drawing %-18s %8.1f env-steps/s %10.0f agent-steps/s" % (r["what"], r["drawing"], r["env_steps_per_s"], r["agent_steps_per_s"]))
    print(prof)
    return 0


if __name__ == "__main__":
    sys.exit(main())
