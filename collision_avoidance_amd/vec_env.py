"""Batched collision-avoidance environment on one MI355X: A independent arenas x N agents.

This is the vectorised form of the reference's Collision_Avoidance_Env (reference
collision_avoidance/envs/collision_avoidence_env.py:23-488): the same reset()/step()/orca_step()
semantics applied to every arena at once by the HIP kernels behind the C ABI of
include/ca_env.h.  PyTorch is used only to hand tensors across (observations out, actions in);
without it the same calls work on numpy arrays through host copies.
"""
import ctypes as C

import numpy as np

from . import _lib, scenarios

try:  # plumbing only
    import torch
except Exception:  # pragma: no cover
    torch = None

_I32_FIELDS = {_lib.FLD_AGENT_DONE, _lib.FLD_ARRIVE_STEP, _lib.FLD_NB_COUNT, _lib.FLD_NB_IDX,
               _lib.FLD_OBST_COUNT, _lib.FLD_OBST_IDX, _lib.FLD_STEP_COUNT, _lib.FLD_ARENA_DONE,
               _lib.FLD_EPISODE, _lib.FLD_REGOAL_COUNT, _lib.FLD_ALAN_ACTION}
_ARENA_FIELDS = {_lib.FLD_STEP_COUNT, _lib.FLD_ARENA_DONE, _lib.FLD_EPISODE}
_F64_FIELDS = {_lib.FLD_GOAL_X, _lib.FLD_GOAL_Y, _lib.FLD_GOAL2_X, _lib.FLD_GOAL2_Y, _lib.FLD_ALAN_WEIGHTS,
               _lib.FLD_ALAN_TIMES}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class VecCollisionAvoidanceEnv:
    """A arenas x N agents advanced per call.

    scenario: "crowd" | "circle" | "doorway" (scenarios.py) or None to leave the state unset.
    params:   dict of ca_config fields; defaults are the reference env's constants.
    use_torch: hand observations/rewards out as torch tensors on the device (zero-copy for the
               observation) and accept device tensors as actions.  False: numpy in/out.
    allow_obst_overflow: the reference's simulator keeps every obstacle edge in range of an agent (env.py:249, 301-318); the
               lists here hold max_obst_neighbors (<= 16).  False (default): an agent with more edges in range makes the step
               calls raise (CA_ERANGE, naming arena and agent); True: the nearest max_obst_neighbors are kept and the event is
               only counted (stats()["obst_overflow"]).
    """

    def __init__(self, n_arenas, n_agents, scenario="crowd", params=None, device=0, seed=0,
                 arena_offset=0, max_obst_neighbors=None, use_torch=None, obstacles="scenario", allow_obst_overflow=False):
        self.L = _lib.load()
        self.A, self.N = int(n_arenas), int(n_agents)
        p = scenarios.env_params()
        if params:
            p.update(params)
        worlds = None   # a world per arena (list of polygon lists) or None: the same polygons for every arena
        if obstacles == "scenario":
            worlds = scenarios.obstacle_worlds(scenario, self.A, n_agents, p["radius"], seed, arena_offset) \
                if scenario is not None else None
            polys = scenarios.obstacles(scenario, n_agents, p["radius"]) if scenario is not None and worlds is None else []
        elif isinstance(obstacles, dict) and "per_arena" in obstacles:
            worlds, polys = list(obstacles["per_arena"]), []
        else:
            polys = list(obstacles or [])
        n_edges = max(sum(len(q) for q in w) for w in worlds) if worlds else sum(len(q) for q in polys)
        if max_obst_neighbors is None:
            max_obst_neighbors = max(1, min(_lib.MAX_OBST_NEIGHBORS, n_edges))
        self.cfg = _lib.Config(n_arenas=self.A, n_agents=self.N, arena_offset=arena_offset, seed=seed,
                               max_obst_neighbors=max_obst_neighbors, **p)
        self.K, self.S = self.cfg.max_neighbors, self.cfg.max_obst_neighbors
        self.device = int(device)
        self.use_torch = (torch is not None and torch.cuda.is_available()) if use_torch is None else bool(use_torch)
        h = C.c_void_p()
        rc = self.L.ca_create(C.byref(self.cfg), self.device, None, C.byref(h))
        _lib.check(self.L, None, rc, "ca_create")
        self.h = h
        if allow_obst_overflow:
            self._call("ca_allow_obstacle_overflow", self.h, 1)
        if self.use_torch:
            # run on PyTorch's current stream (its handle is 0 for the default stream), so that
            # tensor ops and torch.cuda.Event order naturally with the environment's kernels
            torch.cuda.set_device(self.device)
            self._call("ca_set_stream", self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self._obs_t = None
        self.n_actions = 0
        if self.use_torch:
            dev = torch.device("cuda", self.device)
            self._obs_t = torch.zeros((self.A, self.N, _lib.OBS_DIM), dtype=torch.float32, device=dev)
            self._call("ca_bind_obs", self.h, C.c_void_p(self._obs_t.data_ptr()), self._obs_t.numel() * 4)
            self._rew_t = self.field_tensor(_lib.FLD_REWARD)        # views of the library's own buffers: no copies
            self._done_t = self.field_tensor(_lib.FLD_ARENA_DONE)
            self._act_t = torch.zeros((self.A, self.N), dtype=torch.float32, device=dev)
        if worlds is not None:
            self.set_obstacles_per_arena(worlds)
        else:
            self.set_obstacles(polys)
        if scenario is not None:
            self.init_scenario(scenario)

    # ---- plumbing -------------------------------------------------------------------------------
    def _call(self, name, *args):
        rc = getattr(self.L, name)(*args)
        _lib.check(self.L, self.h, rc, name)

    def close(self):
        self.__dict__.pop("_host_bufs", None); self.__dict__.pop("_packed", None); self.__dict__.pop("_packed_act", None)
        for arr in self.__dict__.pop("_host_arrays", []):     # their memory goes with the handle (see host_array)
            try:
                arr.flags.writeable = False
            except Exception:
                pass
        if getattr(self, "h", None):
            self.L.ca_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape_dtype(self, field):
        if field == _lib.FLD_ARENA_STATS:
            return (self.A, 8), np.uint64
        dt = np.int32 if field in _I32_FIELDS else (np.float64 if field in _F64_FIELDS else np.float32)
        if field in _ARENA_FIELDS:
            return (self.A,), dt
        if field == _lib.FLD_NB_IDX:
            return (self.A, max(self.K, 1), self.N), dt
        if field == _lib.FLD_OBST_IDX:
            return (self.A, self.S, self.N), dt
        if field == _lib.FLD_OBS:
            return (self.A, self.N, _lib.OBS_DIM), dt
        if field in (_lib.FLD_ALAN_WEIGHTS, _lib.FLD_ALAN_TIMES):
            return (self.A, self.n_actions, self.N), dt      # device layout; get()/set() present [A, N, n_actions]
        return (self.A, self.N), dt

    def get(self, field, out=None):
        """Host copy of a state field (synchronises the stream).  out: a C-contiguous array of the field's device shape and
        dtype to fill instead of a new one (e.g. host_buffer(field): page-locked, so the copy runs at the link's rate)."""
        shape, dt = self._shape_dtype(field)
        if out is not None and field in (_lib.FLD_ALAN_WEIGHTS, _lib.FLD_ALAN_TIMES):
            raise ValueError("get: out= is not supported for the ALAN weights / times (they are returned transposed, [A, N, n_actions])")
        if out is None:
            out = np.empty(shape, dt)
        elif out.shape != tuple(shape) or out.dtype != np.dtype(dt) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("get: out must be a C-contiguous %s array of shape %s" % (np.dtype(dt).name, tuple(shape)))
        self._call("ca_get", self.h, field, _ptr(out), out.nbytes, 0)
        if field in (_lib.FLD_ALAN_WEIGHTS, _lib.FLD_ALAN_TIMES):
            return np.ascontiguousarray(np.transpose(out, (0, 2, 1)))   # [A, N, n_actions] like ALAN_true.py:75-76
        return out

    def host_array(self, shape, dtype):
        """A numpy array over page-locked, device-visible host memory from the library (ca_host_alloc: no PyTorch involved).
        LIFETIME: the memory belongs to the handle and is released by close() (ca_destroy); the array -- and every view of it,
        which is what host_buffer(), step(copy=False) and step_packed() hand out -- must not be touched after close().
        Copy what has to outlive the environment (np.array(view)); close() marks the arrays it still knows read-only
        as a tripwire for late writers."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        self._call("ca_host_alloc", self.h, max(n, 1), C.byref(ptr))
        buf = (C.c_char * max(n, 1)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self.__dict__.setdefault("_host_arrays", []).append(arr)
        return arr

    def host_buffer(self, field):
        """A persistent page-locked host array for a field (a pageable destination takes the 67-MB observation of 4096 x 64
        agents at ~10 GB/s, a pinned one at the PCIe rate).  Owned by the environment and reused by step(copy=False): valid until
        the next call that writes it -- the reference's own ownership rule (env.py:463-466: the same dict objects every call,
        mutated in place)."""
        bufs = self.__dict__.setdefault("_host_bufs", {})
        if field not in bufs:
            shape, dt = self._shape_dtype(field)
            bufs[field] = self.host_array(shape, dt)
        return bufs[field]

    def step_packed(self, actions=None, with_obs=True, stats=False, autoreset=False, no_done=False):
        """One host-side step in one round trip (ca_step_packed): actions [A,N] (None: the ORCA-only step) in, (obs, rewards, dones,
        step_counts) out as views of ONE page-locked buffer owned by the environment, valid until the next call and never after
        close() (host_array: the buffer is freed with the handle).  The form for one
        environment per worker with results on the host every step (run_rllib.py:77, 108)."""
        an, A = self.A * self.N, self.A
        if "_packed" not in self.__dict__:
            words = an * _lib.OBS_DIM + an + 2 * A
            raw = self.host_array((words,), np.float32)
            self._packed = (raw, raw[:an * _lib.OBS_DIM].reshape(A, self.N, _lib.OBS_DIM), raw[an * _lib.OBS_DIM:an * _lib.OBS_DIM + an].reshape(A, self.N),
                            raw[an * _lib.OBS_DIM + an:an * _lib.OBS_DIM + an + A].view(np.int32), raw[an * _lib.OBS_DIM + an + A:].view(np.int32))
            self._packed_act = self.host_array((A, self.N), np.float32)
        raw, obs, rew, done, steps = self._packed
        flags = (_lib.F_OBS if with_obs else 0) | (_lib.F_STATS if stats else 0) | (_lib.F_AUTORESET if autoreset else 0) | \
                (_lib.F_NODONE if no_done else 0)
        act_ptr = None
        if actions is not None:
            if torch is not None and isinstance(actions, torch.Tensor):
                actions = actions.detach().cpu().numpy()
            self._packed_act[...] = np.asarray(actions, np.float32).reshape(A, self.N)
            act_ptr = self._packed_act.ctypes.data
        self._call("ca_step_packed", self.h, act_ptr, flags, raw.ctypes.data, raw.nbytes)
        return obs, rew, done, steps

    def set(self, field, arr):
        shape, dt = self._shape_dtype(field)
        if field in (_lib.FLD_ALAN_WEIGHTS, _lib.FLD_ALAN_TIMES):
            arr = np.transpose(np.asarray(arr, dt).reshape(self.A, self.N, self.n_actions), (0, 2, 1))
        a = np.ascontiguousarray(np.asarray(arr, dt).reshape(shape))
        self._call("ca_set", self.h, field, _ptr(a), a.nbytes, 0)

    # ---- checkpoint / resume -------------------------------------------------------------------------
    _STATE_FIELDS = ("POS_X", "POS_Y", "VEL_X", "VEL_Y", "PREF_X", "PREF_Y", "GOAL_X", "GOAL_Y", "GOAL2_X", "GOAL2_Y",
                     "AGENT_DONE", "ARRIVE_STEP", "NB_COUNT", "NB_IDX", "OBST_COUNT", "OBST_IDX", "STEP_COUNT", "ARENA_DONE",
                     "EPISODE", "REGOAL_COUNT", "REWARD")

    def get_state(self):
        """Everything the next step depends on, as a dict of host arrays: simulator state, targets, the neighbour lists of
        the last doStep (the observation of a reset reads them, env.py:461-488), the counters that key the random draws
        (episode, re-goal count, step count), the last rewards (an arena frozen by CA_F_FREEZE keeps them) and, after alan_configure,
        the bandit's weights, times and last actions.  The reference never
        serialises its environment (checkpoints exist at the trainer's level only); with counter-based
        draws a restored environment continues bit for bit.  Statistics counters are not part of the state."""
        st = {name: self.get(getattr(_lib, "FLD_" + name)) for name in self._STATE_FIELDS}
        if self.n_actions > 0:
            st["ALAN_WEIGHTS"] = self.get(_lib.FLD_ALAN_WEIGHTS)
            st["ALAN_TIMES"] = self.get(_lib.FLD_ALAN_TIMES)
            st["ALAN_ACTION"] = self.get(_lib.FLD_ALAN_ACTION)
        st["_shape"] = np.array([self.A, self.N, self.K, self.S, self.n_actions], np.int64)
        return st

    def set_state(self, st):
        """Restore a get_state() dict into an environment of the same shape and configuration (seed included: it keys the
        draws); the obstacle tables and the ALAN action set are configuration, not state."""
        shape = [int(v) for v in np.asarray(st["_shape"]).reshape(-1)]
        if shape != [self.A, self.N, self.K, self.S, self.n_actions]:
            raise ValueError("set_state: the state is of an environment of shape %s, this one is %s"
                             % (shape, [self.A, self.N, self.K, self.S, self.n_actions]))
        optional = ("REWARD", "ALAN_ACTION")   # joined in round 5: older snapshots restore without them; anything else
        need = [n for n in self._STATE_FIELDS if n not in optional] + (["ALAN_WEIGHTS", "ALAN_TIMES"] if self.n_actions > 0 else [])
        missing = [n for n in need if n not in st]
        if missing:                            # missing is a truncated / corrupt snapshot: nothing is restored
            raise KeyError("set_state: the snapshot lacks %s" % ", ".join(missing))
        for name in self._STATE_FIELDS:
            if name in st:
                self.set(getattr(_lib, "FLD_" + name), st[name])
        if self.n_actions > 0:
            self.set(_lib.FLD_ALAN_WEIGHTS, st["ALAN_WEIGHTS"])
            self.set(_lib.FLD_ALAN_TIMES, st["ALAN_TIMES"])
            if "ALAN_ACTION" in st:
                self.set(_lib.FLD_ALAN_ACTION, st["ALAN_ACTION"])

    def field_tensor(self, field):
        """Zero-copy torch view of a state field's device buffer (valid until close(); kernels of this handle run on
        torch's current stream, so ordinary stream ordering applies)."""
        if torch is None:
            raise RuntimeError("field_tensor needs PyTorch")
        shape, dt = self._shape_dtype(field)
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        self._call("ca_field_ptr", self.h, field, C.byref(ptr), C.byref(nbytes))

        class _View(object):   # the CUDA array interface, which torch.as_tensor understands
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dt).str, "data": (int(ptr.value), False),
                                      "version": 2, "strides": None}
        t = torch.as_tensor(v, device=torch.device("cuda", self.device))
        assert t.numel() * t.element_size() == nbytes.value
        return t

    def neighbor_lists(self):
        """(count [A,N], idx [A,N,K]) of the ORCA agent neighbours of the last step, nearest first."""
        return self.get(_lib.FLD_NB_COUNT), np.transpose(self.get(_lib.FLD_NB_IDX), (0, 2, 1))[:, :, :self.K]

    def obstacle_neighbor_lists(self):
        return self.get(_lib.FLD_OBST_COUNT), np.transpose(self.get(_lib.FLD_OBST_IDX), (0, 2, 1))

    def set_obstacles(self, polys):
        polys = [np.asarray(q, np.float32).reshape(-1, 2) for q in polys]
        verts = np.ascontiguousarray(np.concatenate(polys) if polys else np.zeros((0, 2), np.float32))
        sizes = np.asarray([len(q) for q in polys], np.int32)
        self._call("ca_set_obstacles", self.h, _ptr(verts), _ptr(sizes), len(polys))

    def set_obstacles_per_arena(self, worlds):
        """A world of its own for every arena (the reference builds one simulator per environment and draws e.g. the
        blocks of ALAN_true.py:359-372 anew for each): worlds[a] = list of polygons of arena a."""
        if len(worlds) != self.A:
            raise ValueError("set_obstacles_per_arena: %d worlds for %d arenas" % (len(worlds), self.A))
        polys = [np.asarray(q, np.float32).reshape(-1, 2) for w in worlds for q in w]
        verts = np.ascontiguousarray(np.concatenate(polys) if polys else np.zeros((0, 2), np.float32))
        sizes = np.asarray([len(q) for q in polys], np.int32)
        counts = np.asarray([len(w) for w in worlds], np.int32)
        self._call("ca_set_obstacles_per_arena", self.h, _ptr(verts), _ptr(sizes), _ptr(counts))

    def obstacle_table(self, arena=0):
        """The processed vertex table of an arena (after the edge cuts of processObstacles): dict(verts [n,2],
        next [n], convex [n]); edge i = verts[i] -> verts[next[i]], the ids of obstacle_neighbor_lists()."""
        n = C.c_int32()
        self._call("ca_get_obstacles_arena", self.h, int(arena), None, None, None, 0, C.byref(n))
        verts, nxt, cvx = np.zeros((n.value, 2), np.float32), np.zeros(n.value, np.int32), np.zeros(n.value, np.int32)
        self._call("ca_get_obstacles_arena", self.h, int(arena), _ptr(verts), _ptr(nxt), _ptr(cvx), n.value, C.byref(n))
        return dict(verts=verts, next=nxt, convex=cvx)

    def init_scenario(self, scenario):
        sid = scenarios.SCENARIO_IDS[scenario] if isinstance(scenario, str) else int(scenario)
        self._call("ca_init_scenario", self.h, sid)

    def sync(self):
        self._call("ca_sync", self.h)

    def stats(self):
        s = _lib.Stats()
        self._call("ca_get_stats", self.h, C.byref(s))
        return s.as_dict()

    def reset_stats(self):
        self._call("ca_reset_stats", self.h)

    def profile(self, period=1):
        """Time the kernel launches of every `period`-th step (start / stop events on the dispatch itself; 0 = off)."""
        self._call("ca_profile", self.h, int(period))

    def profile_read(self):
        """{kernel: (launches, mean ms)} since the last read; synchronises."""
        n = (C.c_int32 * 4)()
        ms = (C.c_float * 4)()
        self._call("ca_profile_read", self.h, n, ms)
        return {k: (n[i], ms[i]) for i, k in enumerate(("nbr_kernel", "step_kernel", "obs_kernel", "reset_kernels"))}

    def launch_info(self):
        v = [C.c_int32() for _ in range(4)]
        self._call("ca_launch_info", self.h, *[C.byref(x) for x in v])
        lanes, roll = C.c_int32(), C.c_int32()
        self._call("ca_solver_info", self.h, C.byref(lanes), C.byref(roll))
        return dict(block=v[0].value, grid=v[1].value, lds_bytes=v[2].value, obs_grid=v[3].value,
                    lanes_per_agent=lanes.value, rollout_one_launch=roll.value)

    # ---- the environment API ----------------------------------------------------------------------
    def _on_device(self, t, dtype):
        """A torch tensor as (contiguous tensor on this handle's GPU, True) or, for a CPU tensor, as
        (numpy array, False): the library dereferences a device pointer as device memory, so a CPU tensor
        must take the host path and a tensor of another GPU is moved here first."""
        if not t.is_cuda:
            return t.detach().to(dtype=dtype).contiguous().numpy(), False
        dev = torch.device("cuda", self.device)
        t = t.detach().to(device=dev, dtype=dtype).contiguous()
        if not self.use_torch:   # the handle runs on a stream of its own: torch's producer must be finished
            torch.cuda.current_stream(self.device).synchronize()
        return t, True

    def _obs_out(self):
        return self._obs_t if self.use_torch else self.get(_lib.FLD_OBS)

    def reset(self, pos_x=None, pos_y=None, with_obs=True):
        """reference reset() (env.py:461-488) for every arena.  pos_x/pos_y [A,N]: explicit new
        positions (numpy or device tensor); default: drawn from the spawn box."""
        flags = _lib.F_OBS if with_obs else 0
        if pos_x is None:
            self._call("ca_reset", self.h, None, None, 0, flags)
        elif torch is not None and isinstance(pos_x, torch.Tensor) and pos_x.is_cuda and \
                isinstance(pos_y, torch.Tensor) and pos_y.is_cuda:
            px, _ = self._on_device(pos_x, torch.float32)
            py, _ = self._on_device(pos_y, torch.float32)
            if px.numel() != self.A * self.N or py.numel() != self.A * self.N:
                raise ValueError("reset: pos_x / pos_y must hold %d x %d values" % (self.A, self.N))
            self._call("ca_reset", self.h, C.c_void_p(px.data_ptr()), C.c_void_p(py.data_ptr()), 1, flags)
            self.sync()
        else:
            if torch is not None and isinstance(pos_x, torch.Tensor):
                pos_x = pos_x.detach().cpu().numpy()
            if torch is not None and isinstance(pos_y, torch.Tensor):
                pos_y = pos_y.detach().cpu().numpy()
            px = np.ascontiguousarray(np.asarray(pos_x, np.float32).reshape(self.A, self.N))
            py = np.ascontiguousarray(np.asarray(pos_y, np.float32).reshape(self.A, self.N))
            self._call("ca_reset", self.h, _ptr(px), _ptr(py), 0, flags)
        return self._obs_out() if with_obs else None

    def reset_masked(self, mask, with_obs=True):
        """reset() for the arenas with mask[a] != 0 only (numpy or device int32 tensor [A])."""
        flags = _lib.F_OBS if with_obs else 0
        on_dev = False
        if torch is not None and isinstance(mask, torch.Tensor):
            mask, on_dev = self._on_device(mask, torch.int32)
        if on_dev:
            if mask.numel() != self.A:
                raise ValueError("reset_masked: mask must hold %d values" % self.A)
            self._call("ca_reset_masked", self.h, C.c_void_p(mask.data_ptr()), 1, flags)
            self.sync()
        else:
            m = np.ascontiguousarray(np.asarray(mask, np.int32).reshape(self.A))
            self._call("ca_reset_masked", self.h, _ptr(m), 0, flags)
        return self._obs_out() if with_obs else None

    def arena_stats(self):
        """Per-arena counters as a dict of [A] arrays (synchronises)."""
        r = self.get(_lib.FLD_ARENA_STATS)
        return dict(episodes=r[:, 0], collisions=r[:, 1], obst_collisions=r[:, 2], goals_reached=r[:, 3],
                    obst_overflow=r[:, 4], sum_reward=r[:, 5].copy().view(np.float64), frozen_steps=r[:, 6],
                    last_episode_steps=(r[:, 7] >> np.uint64(32)).astype(np.int64),
                    last_episode_arrived=(r[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64))

    def step(self, actions, with_obs=True, stats=False, autoreset=False, copy=True):
        """reference step(action) (env.py:367-416) for every arena.
        actions [A,N] heading offsets (rad).  Returns (obs [A,N,64], rewards [A,N], dones [A], {}).
        copy=False (numpy mode): the results land in the environment's own page-locked host buffers (host_buffer) and are
        valid until the next step / reset -- no allocation per step and the device -> host copy at the link's rate."""
        flags = (_lib.F_OBS if with_obs else 0) | (_lib.F_STATS if stats else 0) | \
                (_lib.F_AUTORESET if autoreset else 0)
        if self.use_torch:
            if not isinstance(actions, torch.Tensor):
                actions = torch.as_tensor(np.asarray(actions, np.float32).reshape(self.A, self.N))
            if actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous() and \
                    actions.numel() == self.A * self.N and actions.device.index == self.device:
                src = actions                      # already where and what the kernel reads: no staging copy
            else:
                self._act_t.copy_(actions.reshape(self.A, self.N), non_blocking=True)
                src = self._act_t
            self._call("ca_step", self.h, C.c_void_p(src.data_ptr()), flags)
            return (self._obs_t if with_obs else None), self._rew_t, self._done_t, {}
        a = np.ascontiguousarray(np.asarray(actions, np.float32).reshape(self.A, self.N))
        self._call("ca_step_host", self.h, _ptr(a), flags)
        if not copy:
            return (self.get(_lib.FLD_OBS, self.host_buffer(_lib.FLD_OBS)) if with_obs else None), \
                self.get(_lib.FLD_REWARD, self.host_buffer(_lib.FLD_REWARD)), \
                self.get(_lib.FLD_ARENA_DONE, self.host_buffer(_lib.FLD_ARENA_DONE)), {}
        return (self.get(_lib.FLD_OBS) if with_obs else None), self.get(_lib.FLD_REWARD), \
            self.get(_lib.FLD_ARENA_DONE), {}

    def orca_step(self, with_obs=False, stats=False, no_done=False, autoreset=False, freeze=False):
        """reference orca_step (env.py:447-458 with no_done=True; ALAN_true.py:631-636 + done test).
        freeze: arenas whose episode is over are no longer advanced."""
        flags = (_lib.F_OBS if with_obs else 0) | (_lib.F_STATS if stats else 0) | \
                (_lib.F_NODONE if no_done else 0) | (_lib.F_AUTORESET if autoreset else 0) | \
                (_lib.F_FREEZE if freeze else 0)
        self._call("ca_orca_step", self.h, flags)
        return self._obs_out() if with_obs else None

    def observe(self):
        """reference _get_obs() (env.py:231-277) on the current state."""
        self._call("ca_observe", self.h)
        return self._obs_out()

    def rollout(self, steps, with_obs=False, stats=False, autoreset=False, freeze=False):
        flags = (_lib.F_OBS if with_obs else 0) | (_lib.F_STATS if stats else 0) | \
                (_lib.F_AUTORESET if autoreset else 0) | (_lib.F_FREEZE if freeze else 0)
        self._call("ca_rollout", self.h, int(steps), flags)

    # ---- ALAN online learning (ALAN_true.py:569-628) -----------------------------------------------
    def alan_configure(self, actions, temp=0.2, timewindow=2.0, time_step=1 / 60.):
        """Bandit state of the reference's constructor (ALAN_true.py:31-38, 47-49, 73-76): `actions`
        [n, 2] vectors, weights and times zeroed."""
        a = np.ascontiguousarray(np.asarray(actions, np.float64).reshape(-1, 2))
        self._call("ca_alan_configure", self.h, _ptr(a), a.shape[0], float(temp), float(timewindow), float(time_step))
        self.n_actions = a.shape[0]

    def alan_step(self, u=None, with_obs=False, stats=False, freeze=False):
        """reference online_step (ALAN_true.py:569-628) + step counter + goal test (ALAN:118-121).
        u [A,N] float64: the uniforms behind np.random.choice, one per agent (numpy or device tensor);
        None: the handle's counter-based RNG."""
        flags = (_lib.F_OBS if with_obs else 0) | (_lib.F_STATS if stats else 0) | (_lib.F_FREEZE if freeze else 0)
        if u is None:
            self._call("ca_alan_step", self.h, None, 0, flags)
        elif torch is not None and isinstance(u, torch.Tensor) and u.is_cuda:
            ud, _ = self._on_device(u, torch.float64)
            if ud.numel() != self.A * self.N:
                raise ValueError("alan_step: u must hold %d x %d values" % (self.A, self.N))
            self._call("ca_alan_step", self.h, C.c_void_p(ud.data_ptr()), 1, flags)
            self.sync()
        else:
            if torch is not None and isinstance(u, torch.Tensor):
                u = u.detach().numpy()
            uh = np.ascontiguousarray(np.asarray(u, np.float64).reshape(self.A, self.N))
            self._call("ca_alan_step", self.h, _ptr(uh), 0, flags)
        return self._obs_out() if with_obs else None

    def alan_rollout(self, steps, stats=False, freeze=True):
        """run_sim(mode=1) (ALAN_true.py:106-123) for every arena at once: each arena stops at the end of
        its own episode (freeze) or after `steps` steps."""
        flags = (_lib.F_STATS if stats else 0) | (_lib.F_FREEZE if freeze else 0)
        self._call("ca_alan_rollout", self.h, int(steps), flags)

    def state(self):
        names = dict(pos_x=_lib.FLD_POS_X, pos_y=_lib.FLD_POS_Y, vel_x=_lib.FLD_VEL_X, vel_y=_lib.FLD_VEL_Y,
                     pref_x=_lib.FLD_PREF_X, pref_y=_lib.FLD_PREF_Y, goal_x=_lib.FLD_GOAL_X,
                     goal_y=_lib.FLD_GOAL_Y, agent_done=_lib.FLD_AGENT_DONE,
                     step_count=_lib.FLD_STEP_COUNT, arena_done=_lib.FLD_ARENA_DONE)
        return {k: self.get(v) for k, v in names.items()}
