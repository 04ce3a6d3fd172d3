"""The gym / RLlib entry path of the reference, executed with stub `gym` and `ray` modules (neither is installed here; the
technique tests/golden/make_golden.py uses to import the reference itself):

  collision_avoidance/__init__.py:3-6   register(id='collision_avoidance-v0', entry_point='…envs:Collision_Avoidance_Env')
  run_rllib.py:77-82                    register_env(...); gym.make('collision_avoidance-v0'); .observation_space / .action_space
  collision_avoidence_env.py:23         class Collision_Avoidance_Env(gym.Env, MultiAgentEnv)
  collision_avoidence_env.py:52-53      gym.spaces.Box action / observation spaces

Runs in a child interpreter: the package decides its base classes at import, so the stubs must be in sys.modules before the
first import.  No GPU: the handle construction (`_make`) and `reset` are patched out -- what is under test is the host-side
entry path, not the kernels (tests/test_gpu_parity.py::test_dropin_env_api drives the real class on the card; the `-m gpu`
half of this file runs the same entry path end to end, gym.make -> reset -> step -> seed)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUBS = textwrap.dedent('''
    import importlib, sys, types
    import numpy as np

    class Env(object):
        metadata = {}
    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
    class GymError(Exception):
        pass
    registry = {}
    def register(id, entry_point=None, **kw):
        if id in registry:
            raise GymError("Cannot re-register id: %s" % id)
        registry[id] = (entry_point, kw)
    def make(id, **kw):
        entry_point, kw0 = registry[id]
        mod, _, cls = entry_point.partition(":")
        return getattr(importlib.import_module(mod), cls)(**dict(kw0.get("kwargs") or {}, **kw))
    gym = types.ModuleType("gym"); gym.Env = Env; gym.make = make
    gym.spaces = types.ModuleType("gym.spaces"); gym.spaces.Box = Box
    gym.envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration"); reg.register = register; reg.registry = registry
    gym.envs.registration = reg
    gym.error = types.ModuleType("gym.error"); gym.error.Error = GymError
    class MultiAgentEnv(object):
        pass
    mods = {"gym": gym, "gym.spaces": gym.spaces, "gym.envs": gym.envs, "gym.envs.registration": reg, "gym.error": gym.error}
    ray = types.ModuleType("ray")
    mods["ray"] = ray
    for n in ("ray.rllib", "ray.rllib.env", "ray.rllib.env.multi_agent_env"):
        mods[n] = types.ModuleType(n)
    mods["ray.rllib.env.multi_agent_env"].MultiAgentEnv = MultiAgentEnv
    sys.modules.update(mods)
''')

BODY_CPU = textwrap.dedent('''
    import json
    import collision_avoidance_amd                       # registers the id on import, like the reference package
    from collision_avoidance_amd import envs as E
    out = {"registered": registry.get("collision_avoidance-v0", (None,))[0]}
    out["mro"] = [c.__module__ + "." + c.__name__ for c in E.Collision_Avoidance_Env.__mro__]
    out["is_gym_env"] = issubclass(E.Collision_Avoidance_Env, Env)
    out["is_multi_agent_env"] = issubclass(E.Collision_Avoidance_Env, MultiAgentEnv)
    made = []
    E.Collision_Avoidance_Env._make = lambda self: made.append("make")      # no device in this test
    E.Collision_Avoidance_Env.reset = lambda self: made.append("reset")
    env = gym.make("collision_avoidance-v0")
    out["made"] = made
    out["cls"] = type(env) is E.Collision_Avoidance_Env and isinstance(env, MultiAgentEnv) and isinstance(env, Env)
    out["act"] = [type(env.action_space) is Box, env.action_space.low, env.action_space.high, list(env.action_space.shape)]
    out["obs"] = [type(env.observation_space) is Box, env.observation_space.low, env.observation_space.high,
                  list(env.observation_space.shape)]
    out["numAgents"] = env.numAgents
    env3 = gym.make("collision_avoidance-v0", numAgents=3)
    out["keys3"] = env3._keys
    # a second import path of the package must not fail on the already-registered id, any other registration error must
    try:
        collision_avoidance_amd._register_gym_id(); out["reregister"] = "ok"
    except Exception as e:
        out["reregister"] = repr(e)
    def boom(**kw):
        raise GymError("malformed environment ID")
    reg.register = boom
    try:
        collision_avoidance_amd._register_gym_id(); out["other_error"] = "swallowed"
    except GymError:
        out["other_error"] = "raised"
    print("RESULT " + json.dumps(out))
''')


def _run(body):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", STUBS + body], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_gym_id_resolves_to_the_dropin_class_with_both_bases():
    from math import pi
    out = _run(BODY_CPU)
    assert out["registered"] == "collision_avoidance_amd.envs:Collision_Avoidance_Env"      # collision_avoidance/__init__.py:3-6
    assert out["is_gym_env"] and out["is_multi_agent_env"] and out["cls"]                    # env.py:23
    assert out["mro"][0].endswith("envs.Collision_Avoidance_Env")
    assert [m.split(".")[-1] for m in out["mro"][1:3]] == ["Env", "MultiAgentEnv"]         # the reference's base order
    assert out["made"][:2] == ["make", "reset"]                                              # env.py:74: the constructor resets
    assert out["act"] == [True, -pi, pi, [1]]                                                # env.py:52
    assert out["obs"] == [True, -1.5, 1.5, [64]]                                             # env.py:53
    assert out["numAgents"] == 10 and out["keys3"] == ["agent_0", "agent_1", "agent_2"]      # env.py:24, 275
    assert out["reregister"] == "ok" and out["other_error"] == "raised"


def test_package_imports_without_gym_or_ray():
    """Neither is installed in this image: the package and the drop-in class import, nothing is registered, the class is a plain
    object subclass with the stand-in Box."""
    code = ("import sys, json; import collision_avoidance_amd as P; from collision_avoidance_amd import envs as E; "
            "print('RESULT ' + json.dumps({'reg': P._register_gym_id(), 'bases': [b.__name__ for b in E.Collision_Avoidance_Env.__bases__], "
            "'gym': 'gym' in sys.modules}))")
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out == {"reg": False, "bases": ["object"], "gym": False}


BODY_GPU = textwrap.dedent('''
    import json
    import numpy as np
    import collision_avoidance_amd
    single_env = gym.make("collision_avoidance-v0")            # run_rllib.py:82
    out = {"obs_shape": list(single_env.observation_space.shape), "act_shape": list(single_env.action_space.shape)}
    obs = single_env.reset()
    out["keys"] = sorted(obs.keys()) == sorted("agent_%d" % i for i in range(10))
    act = {"agent_%d" % i: np.array([0.1 * i - 0.5], np.float32) for i in range(10)}
    for _ in range(5):
        o, r, d, info = single_env.step(act)
    out["row"] = len(o["agent_0"]); out["all"] = d["__all__"]; out["rew"] = [r["agent_%d" % i] for i in range(10)]
    # seed(): [seed] back (env.py:494-496), the whole state carried over -- also the counters that key later draws
    from collision_avoidance_amd import _lib
    before = single_env.vec.get_state()
    out["seed_ret"] = single_env.seed(7)
    after = single_env.vec.get_state()
    out["state_kept"] = all(np.array_equal(np.asarray(before[k]).view(np.uint8), np.asarray(after[k]).view(np.uint8)) for k in before)
    out["fields"] = sorted(before.keys())
    out["step_count"] = single_env.step_count
    o2, r2, d2, _ = single_env.step(act)
    out["row2"] = len(o2["agent_3"])
    single_env.close()
    print("RESULT " + json.dumps(out))
''')


@pytest.mark.gpu
def test_gym_make_reset_step_seed_on_the_card():
    out = _run(BODY_GPU)
    assert out["obs_shape"] == [64] and out["act_shape"] == [1] and out["keys"] and out["row"] == 64 and out["row2"] == 64
    assert out["all"] is False and len(out["rew"]) == 10 and all(abs(v) <= 1.0 + 1e-6 for v in out["rew"])
    assert out["seed_ret"] == [7] and out["state_kept"] and out["step_count"] == 5
    assert {"EPISODE", "REGOAL_COUNT", "ARRIVE_STEP", "STEP_COUNT", "NB_IDX"} <= set(out["fields"])
