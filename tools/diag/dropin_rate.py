#!/usr/bin/env python3
"""Diagnostic: the reference's own usage -- ONE environment through the drop-in class (collision_avoidence_env.py:23, main :570-573):
env-steps/s of step(action_dict) and orca_step(), and how a step's time splits between the library and the Python dictionaries."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from collision_avoidance_amd.envs import Collision_Avoidance_Env
for n in (10, 8):
    env = Collision_Avoidance_Env(n)
    rng = np.random.RandomState(0)
    acts = [{'agent_' + str(i): np.array([rng.uniform(-0.5, 0.5)]) for i in range(n)} for _ in range(8)]
    for i in range(200):
        env.step(acts[i % 8])
    N = 2000
    t0 = time.perf_counter()
    for i in range(N):
        obs, rew, dones, infos = env.step(acts[i % 8])
        if dones['__all__']:
            env.reset()
    dt = (time.perf_counter() - t0) / N
    a = np.zeros((1, n), np.float32)
    t0 = time.perf_counter()
    for i in range(N):
        env.vec.step(a)
    dv = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for i in range(N):
        env.orca_step()
    do = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for i in range(N):
        env.vec.step_packed(a)
    dp = (time.perf_counter() - t0) / N
    print("drop-in env, %d agents: step(dict) %.1f us = %.0f env-steps/s (%.0f agent-steps/s), of which the library's one round trip "
          "(ca_step_packed) %.1f us; the same step as ca_step_host + three ca_get (four synchronisations, rounds 1-4) %.1f us; "
          "orca_step() %.1f us" % (n, dt * 1e6, 1 / dt, n / dt, dp * 1e6, dv * 1e6, do * 1e6))
    env.close()
