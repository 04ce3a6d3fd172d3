"""ALAN evaluation metrics and a drop-in for `Collision_Avoidance_Sim` (reference
collision_avoidance/ALAN/ALAN_true.py:10-172, 547-636) on top of the batched HIP environment.

The simulator runs one arena; what the reference computes per agent in Python around
`sim.doStep()` -- softmax action selection, the bandit update, the goal test -- runs on the device
(`ca_alan_step`).  The metrics are plain numpy restatements of ALAN_true.py:125-131 and :161-172.
"""
import numpy as np

from . import scenarios

DEFAULT_ACTIONS = [(1, 0), (0.70711, 0.70711), (0, 1), (-0.70711, 0.70711), (-1, 0), (-0.70711, -0.70711),
                   (0, -1), (0.70711, -0.70711)]                                     # ALAN_true.py:31-38


def agents_time(arrive_step, agent_done, time_step, max_step):
    """ALAN_true.py:62, 559: arrival time = step_count * timeStep when the goal test fired, else the cap."""
    arrive_step = np.asarray(arrive_step)
    t = arrive_step.astype(np.float64) * time_step
    return np.where(np.asarray(agent_done) != 0, t, max_step * time_step)


def ttime(times):
    """ALAN_true.py:127-130: mean + 3 standard deviations of the arrival times."""
    times = np.asarray(times, np.float64)
    return float(np.average(times) + 3 * np.std(times, 0))


def min_ttime(start_xy, goal_xy, max_speed=1.0):
    """ALAN_true.py:161-172: the same statistic of the straight-line travel times."""
    start_xy, goal_xy = np.asarray(start_xy, np.float64), np.asarray(goal_xy, np.float64)
    d = np.sqrt((goal_xy[:, 0] - start_xy[:, 0]) ** 2 + (goal_xy[:, 1] - start_xy[:, 1]) ** 2)
    return ttime(max_speed * d)


class Collision_Avoidance_Sim(object):
    """Same constructor and run_sim()/reset() contract as ALAN_true.py:10-131 (no Tk window).

    n_arenas > 1 runs that many independent episodes of the scenario side by side (arena g of a handle
    is keyed by (seed, g)); run_sim() then returns arrays with one entry per arena."""

    POLL = 256   # steps enqueued between two looks at the arena_done flags

    def __init__(self, numAgents=50, scenario="crowd", online_actions=None, visualize=False, device=0, seed=0,
                 n_arenas=1):
        self.numAgents, self.scenario = numAgents, scenario
        self.timeStep, self.maxSpeed, self.radius = 1 / 60., 1, 0.5
        self.gamma, self.timewindow, self.online_temp = 0.6, 2, 0.2                # ALAN_true.py:47-49
        self.default_online_actions = list(DEFAULT_ACTIONS)
        self.max_step = int((10 / self.timeStep) * numAgents)                       # ALAN_true.py:59
        self._device, self._seed, self.n_arenas = device, seed, int(n_arenas)
        self._resets = 0
        self.vec = None
        self.reset(online_actions)

    def reset(self, online_actions=None):
        from . import _lib
        from .vec_env import VecCollisionAvoidanceEnv
        self.online_actions = self.default_online_actions if online_actions is None else list(online_actions)
        if self.vec is not None:
            self.vec.close()
        p = scenarios.alan_params(self.numAgents, self.scenario)
        self.envsize = scenarios.envsize(self.scenario, self.numAgents)
        # every reset() is a new random world (ALAN_true.py:133-139 re-runs _init_world): the arenas of round r
        # are the global arenas [r * n_arenas, (r + 1) * n_arenas) of this seed
        self.vec = VecCollisionAvoidanceEnv(self.n_arenas, self.numAgents, scenario=self.scenario, params=p,
                                            device=self._device, seed=self._seed, use_torch=False,
                                            arena_offset=self._resets * self.n_arenas)
        self._resets += 1
        self.vec.alan_configure(self.online_actions, self.online_temp, self.timewindow, self.timeStep)
        self.step_count, self.TTime = 0, 0
        start = np.stack([self.vec.get(_lib.FLD_POS_X), self.vec.get(_lib.FLD_POS_Y)], -1)
        goal = np.stack([self.vec.get(_lib.FLD_GOAL_X), self.vec.get(_lib.FLD_GOAL_Y)], -1)
        self._min_ttimes = np.array([min_ttime(start[a], goal[a], self.maxSpeed) for a in range(self.n_arenas)])
        self.min_TTime = float(self._min_ttimes[0])

    def run_sim(self, mode=1):
        """mode 1: ALAN online learning, 0: plain ORCA (ALAN_true.py:106-131).
        Returns (success, total_time, TTime, min_TTime); arrays of length n_arenas when n_arenas > 1.
        Every arena stops at its own last arrival (or at max_step) on the device, like the `break` at
        ALAN_true.py:121-123; the host only looks at the flags every POLL steps."""
        from . import _lib
        left = self.max_step - self.step_count
        while left > 0:
            n = min(self.POLL, left)
            if mode == 0:
                self.vec.rollout(n, freeze=True)
            else:
                self.vec.alan_rollout(n, freeze=True)
            left -= n
            if self.vec.get(_lib.FLD_ARENA_DONE).all():
                break
        steps = self.vec.get(_lib.FLD_STEP_COUNT)
        done = self.vec.get(_lib.FLD_AGENT_DONE)
        arrive = self.vec.get(_lib.FLD_ARRIVE_STEP)
        self.step_count = int(steps[0])
        self.agents_done = [int(d) for d in done[0]]
        times = agents_time(arrive, done, self.timeStep, self.max_step)
        self.agents_time = list(times[0])
        success = done.all(axis=1)
        ttimes = np.array([ttime(times[a]) for a in range(self.n_arenas)])
        self.TTime = float(ttimes[0])
        if self.n_arenas == 1:
            return bool(success[0]), self.step_count * self.timeStep, self.TTime, self.min_TTime
        return success, steps * self.timeStep, ttimes, self._min_ttimes
