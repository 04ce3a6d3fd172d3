"""Vector-environment front ends for trainers.

The reference trains through RLlib's `MultiAgentEnv` with one simulator per worker
(/root/reference/run_rllib.py:77, 108-121; collision_avoidance/envs/collision_avoidence_env.py:23,
367-416, 461-488).  Here one handle advances A arenas x N agents per call, and these two classes
present that batch in the shapes trainers consume:

* `AgentVectorEnv`      -- gym.vector-style: every agent of every arena is one sub-environment of a
                           shared policy (all agents have the same spaces, env.py:52-53); arrays in,
                           arrays out, arenas that finish are reset inside the same call.
* `MultiAgentVectorEnv` -- RLlib-shaped: A multi-agent sub-environments with the reference's
                           'agent_<i>' dictionaries; implements both the `VectorEnv` calls
                           (`vector_reset / reset_at / vector_step`) and the `BaseEnv` calls
                           (`poll / send_actions / try_reset`).

Neither needs gym or ray to be installed; when they are importable the classes can be registered as
they are (duck typing).  What the reference leaves unfinished is finished here: per-agent `dones`
(env.py:463-474 creates them and never sets them), `infos` carrying the collision statistics, and the
distinction between an episode that ended because everybody arrived and one cut by the step cap
(env.py:408-410).
"""
from math import pi

import numpy as np

from . import _lib

try:  # plumbing only
    import torch
except Exception:  # pragma: no cover
    torch = None


class BoxSpace(object):
    """gym.spaces.Box look-alike (env.py:52-53) for installations without gym."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = float(low), float(high), tuple(shape), dtype

    def sample(self):
        return np.random.uniform(self.low, self.high, self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "BoxSpace(%g, %g, %r)" % (self.low, self.high, self.shape)


def _spaces(vec, batch):
    nd = float(vec.cfg.neighbor_dist)
    single_obs = BoxSpace(-nd, nd, (_lib.OBS_DIM,))          # env.py:53
    single_act = BoxSpace(-pi, pi, (1,))                     # env.py:52
    return single_obs, single_act, BoxSpace(-nd, nd, (batch, _lib.OBS_DIM)), BoxSpace(-pi, pi, (batch, 1))


def _episode_info(st, ended, n_agents):
    """Per finished arena: length, arrivals, whether the cap cut it (env.py:408-410)."""
    idx = np.nonzero(ended)[0]
    steps = st["last_episode_steps"][idx]
    arrived = st["last_episode_arrived"][idx]
    return dict(arena=idx, length=steps, arrived=arrived, truncated=arrived < n_agents)


class AgentVectorEnv(object):
    """gym.vector-style view of a VecCollisionAvoidanceEnv: num_envs = A * N agents.

    step(actions) -> (obs [A*N, 64], rewards [A*N], dones [A*N], infos); an agent's `done` is its
    arena's episode end (everybody arrived, or the step cap), and such an arena is reset inside the
    call, so the observation returned for it is the first one of its next episode -- the
    gym.vector auto-reset convention.  new_step_api=True returns (obs, rewards, terminated,
    truncated, infos) instead.  Tensors stay on the device when the underlying env uses torch.

    infos (arrays over arenas unless noted):
      "agent_arrived" [A*N]  the per-agent flag the reference never reports (its gym_dones['agent_i'])
      "collisions", "obst_collisions", "goals_reached"  counted during this step, per arena
      "episode"  dict(arena, length, arrived, truncated) for the arenas that finished in this step
    """

    def __init__(self, vec, new_step_api=False, collect_stats=True):
        self.vec, self.A, self.N = vec, vec.A, vec.N
        self.num_envs = self.A * self.N
        self.new_step_api, self.collect_stats = bool(new_step_api), bool(collect_stats)
        (self.single_observation_space, self.single_action_space,
         self.observation_space, self.action_space) = _spaces(vec, self.num_envs)
        self.is_vector_env = True
        self._prev = None

    def _flat(self, x, tail=()):
        return x.reshape((self.num_envs,) + tuple(tail))

    def reset(self, seed=None, options=None):
        obs = self._flat(self.vec.reset(), (_lib.OBS_DIM,))
        self._prev = self.vec.arena_stats() if self.collect_stats else None
        return (obs, {}) if self.new_step_api else obs

    def step(self, actions):
        a = actions.reshape(self.A, self.N) if hasattr(actions, "reshape") else np.asarray(actions, np.float32).reshape(self.A, self.N)
        obs, rew, done, _ = self.vec.step(a, with_obs=True, stats=self.collect_stats, autoreset=True)
        if torch is not None and isinstance(done, torch.Tensor):
            done_agents = done.reshape(self.A, 1).expand(self.A, self.N).reshape(self.num_envs) != 0
            ended = done.cpu().numpy() != 0 if self.collect_stats else None
        else:
            ended = np.asarray(done) != 0
            done_agents = np.repeat(ended, self.N)
        infos = {}
        trunc_agents = None
        if self.collect_stats:
            st = self.vec.arena_stats()
            for k in ("collisions", "obst_collisions", "goals_reached"):
                infos[k] = (st[k] - self._prev[k]).astype(np.int64)
            self._prev = st
            infos["agent_arrived"] = self.vec.get(_lib.FLD_AGENT_DONE).reshape(self.num_envs) != 0
            infos["episode"] = _episode_info(st, ended, self.N)
            trunc = np.zeros(self.A, bool)
            trunc[infos["episode"]["arena"]] = infos["episode"]["truncated"]
            trunc_agents = np.repeat(trunc, self.N)
        obs, rew = self._flat(obs, (_lib.OBS_DIM,)), self._flat(rew)
        if not self.new_step_api:
            return obs, rew, done_agents, infos
        if trunc_agents is None:
            raise RuntimeError("new_step_api needs collect_stats=True to tell truncation from termination")
        if torch is not None and isinstance(done_agents, torch.Tensor):
            t = torch.as_tensor(trunc_agents, device=done_agents.device)
            return obs, rew, done_agents & ~t, t, infos
        return obs, rew, done_agents & ~trunc_agents, trunc_agents, infos

    def close(self):
        self.vec.close()


class MultiAgentVectorEnv(object):
    """A vector of A multi-agent sub-environments with the reference's dictionaries
    ({'agent_<i>': ...}, '__all__' in dones; env.py:367-416, 461-488).

    per_agent_dones=False : dones['agent_i'] stays False like the reference (env.py:470).
    per_agent_dones=True  : an agent is reported done once, in the step it arrives, and afterwards left
                            out of the observation / reward dictionaries until its arena resets (what
                            RLlib expects of a finished agent); it keeps walking to its second target
                            with a zero heading offset, as a policy-free ORCA agent.
    infos[env]['__common__'] carries the arena's collision counters of the step (common_info=True).
    """

    def __init__(self, vec, per_agent_dones=True, common_info=True):
        self.vec, self.num_envs, self.N = vec, vec.A, vec.N
        self.per_agent_dones, self.common_info = bool(per_agent_dones), bool(common_info)
        self.agent_ids = ['agent_%d' % i for i in range(self.N)]
        self.single_observation_space, self.single_action_space, _, _ = _spaces(vec, 1)
        self.observation_space, self.action_space = self.single_observation_space, self.single_action_space
        self._reported = np.zeros((self.num_envs, self.N), bool)   # agents already reported done
        self._prev = None
        self._pending = None      # BaseEnv: results waiting for poll()
        self._obs = None

    # ---- helpers -----------------------------------------------------------------------------------
    def _host(self, x):
        return x.detach().cpu().numpy() if torch is not None and isinstance(x, torch.Tensor) else np.asarray(x)

    def _obs_dict(self, obs, e):
        live = ~self._reported[e]
        return {self.agent_ids[i]: obs[e, i] for i in range(self.N) if live[i]}

    def _actions(self, action_dicts):
        act = np.zeros((self.num_envs, self.N), np.float32)
        items = action_dicts.items() if isinstance(action_dicts, dict) else enumerate(action_dicts)
        for e, d in items:
            for i, aid in enumerate(self.agent_ids):
                if self._reported[e, i]:
                    continue
                act[e, i] = float(np.asarray(d[aid]).reshape(-1)[0])   # KeyError like env.py:373
        return act

    # ---- RLlib VectorEnv calls ------------------------------------------------------------------------
    def vector_reset(self):
        obs = self._host(self.vec.reset())
        self._reported[:] = False
        self._prev = self.vec.arena_stats()
        self._obs = obs
        return [self._obs_dict(obs, e) for e in range(self.num_envs)]

    def reset_at(self, index):
        mask = np.zeros(self.num_envs, np.int32)
        mask[index] = 1
        obs = self._host(self.vec.reset_masked(mask))
        self._reported[index] = False
        self._obs = obs
        return self._obs_dict(obs, index)

    def vector_step(self, actions):
        act = self._actions(actions)
        obs, rew, done, _ = self.vec.step(act, with_obs=True, stats=True, autoreset=False)
        obs, rew, done = self._host(obs), self._host(rew), self._host(done) != 0
        arrived = self.vec.get(_lib.FLD_AGENT_DONE) != 0
        st = self.vec.arena_stats()
        obs_b, rew_b, done_b, info_b = [], [], [], []
        for e in range(self.num_envs):
            was = self._reported[e].copy()
            newly = arrived[e] & ~was if self.per_agent_dones else np.zeros(self.N, bool)
            dd = {'__all__': bool(done[e])}
            od, rd, idd = {}, {}, {}
            for i, aid in enumerate(self.agent_ids):
                if was[i]:
                    continue
                od[aid], rd[aid], idd[aid] = obs[e, i], float(rew[e, i]), {}
                dd[aid] = bool(newly[i]) or (self.per_agent_dones and bool(done[e]))
            if self.per_agent_dones:
                self._reported[e] |= newly
            if self.common_info:
                idd['__common__'] = dict(
                    collisions=int(st["collisions"][e] - self._prev["collisions"][e]),
                    obst_collisions=int(st["obst_collisions"][e] - self._prev["obst_collisions"][e]),
                    goals_reached=int(st["goals_reached"][e] - self._prev["goals_reached"][e]),
                    truncated=bool(done[e]) and not bool(arrived[e].all()))
            obs_b.append(od); rew_b.append(rd); done_b.append(dd); info_b.append(idd)
        self._prev, self._obs = st, obs
        return obs_b, rew_b, done_b, info_b

    def get_sub_environments(self):
        return []            # the sub-environments are slices of one device batch, not Python objects

    get_unwrapped = get_sub_environments

    # ---- RLlib BaseEnv calls -----------------------------------------------------------------------
    def poll(self):
        if self._pending is None:
            obs = self.vector_reset()
            self._pending = (obs, [{a: 0.0 for a in o} for o in obs],
                             [dict({a: False for a in o}, __all__=False) for o in obs], [{a: {} for a in o} for o in obs])
        obs, rew, done, info = self._pending
        self._pending = None
        ids = range(self.num_envs)
        return ({e: obs[e] for e in ids}, {e: rew[e] for e in ids}, {e: done[e] for e in ids},
                {e: info[e] for e in ids}, {})

    def send_actions(self, action_dict):
        self._pending = self.vector_step(action_dict)

    def try_reset(self, env_id=None):
        if env_id is None:
            return {e: o for e, o in enumerate(self.vector_reset())}
        return {env_id: self.reset_at(env_id)}

    def stop(self):
        self.vec.close()

    close = stop
