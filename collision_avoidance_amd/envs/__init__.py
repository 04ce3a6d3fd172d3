"""Drop-in for `collision_avoidance.envs.Collision_Avoidance_Env` (reference
collision_avoidance/envs/collision_avoidence_env.py:23; exported at envs/__init__.py:1 and
registered as 'collision_avoidance-v0' at collision_avoidance/__init__.py:3-6).

Same constructor, same reset()/step()/orca_step()/seed()/render()/close() signatures, the same
'agent_<i>' dictionaries -- one arena of the batched HIP environment underneath.  gym / RLlib base
classes are used when they are importable and are not required.
"""
from math import pi

import numpy as np

from .. import _lib
from ..vec_env import VecCollisionAvoidanceEnv

try:  # optional: neither is needed to run the environment
    import gym as _gym
    _EnvBase = _gym.Env
except Exception:  # pragma: no cover
    _gym = None
    _EnvBase = object
try:
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _MultiAgentEnv
except Exception:  # pragma: no cover
    _MultiAgentEnv = object


class Box(object):
    """Minimal stand-in for gym.spaces.Box when gym is absent (env.py:52-53)."""

    def __init__(self, low, high, shape):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.float32

    def sample(self):
        return np.random.uniform(self.low, self.high, self.shape).astype(np.float32)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


_bases = (_EnvBase,) if _MultiAgentEnv is object or _EnvBase is _MultiAgentEnv else (_EnvBase, _MultiAgentEnv)


class Collision_Avoidance_Env(*_bases):
    metadata = {'render.modes': ['human']}

    def __init__(self, numAgents=10, device=0, seed=0):
        # constants of env.py:27-44
        self.timeStep = 1 / 60.
        self.neighborDist = 1.5
        self.maxNeighbors = 5
        self.timeHorizon = 1.5
        self.radius = 0.5
        self.maxSpeed = 1
        self.laser_num = 16
        self.circle_approx_num = 8
        self.numAgents = numAgents
        self.envsize = 10
        self.max_step = 1000
        box = _gym.spaces.Box if _gym is not None and hasattr(_gym, "spaces") else Box
        self.action_space = box(low=-pi, high=pi, shape=(1,))                                   # env.py:52
        self.observation_space = box(low=-self.neighborDist, high=self.neighborDist,
                                     shape=(self.laser_num * 4,))                               # env.py:53
        self._device, self._seed = device, seed
        self._keys = ['agent_' + str(i) for i in range(numAgents)]                              # env.py:275, 373, 400
        self._make()
        self.reset()                                                                            # env.py:74

    def _make(self):
        self.vec = VecCollisionAvoidanceEnv(1, self.numAgents, scenario="doorway", device=self._device,
                                            seed=self._seed, use_torch=False)

    @property
    def step_count(self):
        return int(self.vec.get(_lib.FLD_STEP_COUNT)[0])

    @property
    def agents_done(self):
        return [int(v) for v in self.vec.get(_lib.FLD_AGENT_DONE)[0]]

    def _fill_obs(self, obs):
        rows = obs[0].tolist()      # 64 Python floats per agent, as the reference's lists (env.py:275)
        for k, row in zip(self._keys, rows):
            self.gym_obs[k] = row
        return self.gym_obs

    def reset(self):
        # env.py:461-488: the four dictionaries are re-created, then mutated in place by step()
        self.gym_obs, self.gym_rewards, self.gym_dones, self.gym_infos = {}, {}, {'__all__': False}, {}
        for i in range(self.numAgents):
            self.gym_rewards['agent_' + str(i)] = 0
            self.gym_dones['agent_' + str(i)] = False
            self.gym_infos['agent_' + str(i)] = {}
        return self._fill_obs(self.vec.reset())

    def step(self, action):
        # env.py:367-416.  A missing agent key raises KeyError like the reference (env.py:373).
        act = [float(np.asarray(action[k]).reshape(-1)[0]) for k in self._keys]
        obs, rew, done, _ = self.vec.step_packed(act)      # one round trip: actions in, obs | reward | done out (ca_step_packed)
        for k, r in zip(self._keys, rew[0].tolist()):
            self.gym_rewards[k] = r
        self.gym_dones['__all__'] = bool(done[0])
        self._fill_obs(obs)
        return self.gym_obs, self.gym_rewards, self.gym_dones, self.gym_infos

    def orca_step(self, action=None):
        # env.py:447-458 (no done test, no step counter); returns None, updates self.gym_obs
        self._fill_obs(self.vec.step_packed(None, no_done=True)[0])

    def seed(self, seed=None):
        # env.py:494-496 returns [seed]; here the seed also keys the spawn draws of later resets.  The handle is rebuilt
        # with the new seed and the WHOLE state carried over (get_state / set_state: simulator state, targets, lists, and
        # the counters that key the draws -- episode, re-goal count, arrival steps, step count)
        self._seed = 0 if seed is None else int(seed)
        state = self.vec.get_state()
        self.vec.close()
        self._make()
        self.vec.set_state(state)
        return [seed]

    def render(self, mode='human'):
        """The reference draws a Tk window (env.py:491, 504-567); there is no display here."""
        return None

    def close(self):
        """The reference blocks on input() (env.py:499-501); this releases the device buffers."""
        self.vec.close()


CollisionAvoidanceEnv = Collision_Avoidance_Env
__all__ = ["Collision_Avoidance_Env", "CollisionAvoidanceEnv", "Box"]
