#!/usr/bin/env python3
"""Hand-over helpers around the ALAN action-space trainer (reference Train_ALAN_action_space.py; SURVEY.md section 2 rows
8-9: out of the hot path's scope, so not part of the product package): reading / writing the trainer's `.act` files and
the trainer's `evaluate_action` as one batched run of the HIP environment."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_actions(path):
    """Read an action set written by the reference's trainer (`f.write(str(actions))`,
    Train_ALAN_action_space.py:150-153: the repr of a list of (x, y) tuples, e.g. ALAN/crowd_actions.act)."""
    import ast
    with open(path) as f:
        acts = ast.literal_eval(f.read().strip())
    out = [(float(a[0]), float(a[1])) for a in acts]
    if not out or len(out) > 32:
        raise ValueError("%s: %d actions (supported: 1..32)" % (path, len(out)))
    return out


def save_actions(path, actions):
    """The same format, for hand-over to the reference's own simulator."""
    with open(path, "w") as f:
        f.write(str([(float(a[0]), float(a[1])) for a in actions]))


def evaluate_actions(actions, numAgents=50, scenario="crowd", num=3, mode=1, device=0, seed=0):
    """Mean TTime of an action set over `num` random worlds -- what MCMC_trainer.evaluate_action
    (Train_ALAN_action_space.py:53-66) computes with `num` reset()/run_sim() rounds in sequence -- as ONE
    batched run: the `num` episodes are the arenas of one handle, each arena a world of its own (in the "blocks"
    scenario every arena draws its own four blocks, like every reset() of the reference, ALAN_true.py:92-100, 359-372).
    Returns (mean TTime, successes)."""
    from collision_avoidance_amd.alan import Collision_Avoidance_Sim
    sim = Collision_Avoidance_Sim(numAgents=numAgents, scenario=scenario, online_actions=actions, device=device,
                                  seed=seed, n_arenas=num)
    ok, _, tt, _ = sim.run_sim(mode)
    sim.vec.close()
    return float(np.mean(tt)), int(np.sum(ok))
