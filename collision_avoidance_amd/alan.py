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
    """Same constructor and run_sim()/reset() contract as ALAN_true.py:10-131 (no Tk window)."""

    def __init__(self, numAgents=50, scenario="crowd", online_actions=None, visualize=False, device=0, seed=0):
        self.numAgents, self.scenario = numAgents, scenario
        self.timeStep, self.maxSpeed, self.radius = 1 / 60., 1, 0.5
        self.gamma, self.timewindow, self.online_temp = 0.6, 2, 0.2                # ALAN_true.py:47-49
        self.default_online_actions = list(DEFAULT_ACTIONS)
        self.max_step = int((10 / self.timeStep) * numAgents)                       # ALAN_true.py:59
        self._device, self._seed = device, seed
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
        self.vec = VecCollisionAvoidanceEnv(1, self.numAgents, scenario=self.scenario, params=p,
                                            device=self._device, seed=self._seed, use_torch=False)
        self.vec.alan_configure(self.online_actions, self.online_temp, self.timewindow)
        self.step_count, self.TTime = 0, 0
        start = np.stack([self.vec.get(_lib.FLD_POS_X)[0], self.vec.get(_lib.FLD_POS_Y)[0]], 1)
        goal = np.stack([self.vec.get(_lib.FLD_GOAL_X)[0], self.vec.get(_lib.FLD_GOAL_Y)[0]], 1)
        self.min_TTime = min_ttime(start, goal, self.maxSpeed)

    def run_sim(self, mode=1):
        """mode 1: ALAN online learning, 0: plain ORCA (ALAN_true.py:106-131).
        Returns (success, total_time, TTime, min_TTime)."""
        from . import _lib
        success = False
        for _ in range(self.max_step):
            if mode == 0:
                self.vec.orca_step()
            else:
                self.vec.alan_step()
            self.step_count += 1
            poll = self.step_count % 16 == 0 or self.step_count == self.max_step
            if poll and int(self.vec.get(_lib.FLD_ARENA_DONE)[0]):
                success = bool(self.vec.get(_lib.FLD_AGENT_DONE)[0].all())
                break
        done = self.vec.get(_lib.FLD_AGENT_DONE)[0]
        self.agents_done = [int(d) for d in done]
        self.agents_time = list(agents_time(self.vec.get(_lib.FLD_ARRIVE_STEP)[0], done, self.timeStep, self.max_step))
        success = bool(done.all())
        # the device keeps stepping finished agents between polls; the episode length is the last arrival
        total_steps = int(self.vec.get(_lib.FLD_ARRIVE_STEP)[0].max()) if success else self.step_count
        self.TTime = ttime(self.agents_time)
        return success, total_steps * self.timeStep, self.TTime, self.min_TTime
