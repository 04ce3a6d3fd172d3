"""`rvo2`-shaped module over the CPU oracle.  TEST INFRASTRUCTURE ONLY (this container only).

The reference imports the third-party Cython module `rvo2` (env.py:16, ALAN_true.py:6), which
is absent here.  tests/golden/make_golden.py injects this shim into sys.modules so that the
reference's OWN env code (step / reset / _get_obs / done_test) can be executed on top of the
oracle's ORCA and its outputs recorded as golden vectors.  Method names and argument order are
those the reference calls ("lower boundary").
"""
from . import oracle as _o


class PyRVOSimulator:
    def __init__(self, timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius,
                 maxSpeed, velocity=(0.0, 0.0)):
        self._L = _o.lib()
        self._h = self._L.orc_sim_create(timeStep, neighborDist, int(maxNeighbors), timeHorizon,
                                         timeHorizonObst, radius, maxSpeed, velocity[0], velocity[1])
        self._defaults = (neighborDist, int(maxNeighbors), timeHorizon, timeHorizonObst, radius,
                          maxSpeed, tuple(velocity))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_sim_destroy(self._h)
            self._h = None

    # ---- construction ----
    def addAgent(self, pos, neighborDist=None, maxNeighbors=None, timeHorizon=None,
                 timeHorizonObst=None, radius=None, maxSpeed=None, velocity=None):
        d = self._defaults
        args = [neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed, velocity]
        args = [d[i] if a is None else a for i, a in enumerate(args)]
        i = self._L.orc_sim_add_agent(self._h, pos[0], pos[1], args[0], int(args[1]), args[2], args[3],
                                      args[4], args[5], args[6][0], args[6][1])
        if i < 0:
            raise RuntimeError("Error adding agent to RVO simulation")
        return i

    def addObstacle(self, vertices):
        import numpy as np
        v = np.ascontiguousarray(np.asarray(vertices, np.float32).reshape(-1, 2))
        i = self._L.orc_sim_add_obstacle(self._h, v.ctypes.data, v.shape[0])
        if i < 0:
            raise RuntimeError("Error adding obstacle to RVO simulation")
        return i

    def processObstacles(self):
        self._L.orc_sim_process_obstacles(self._h)

    def doStep(self):
        self._L.orc_sim_do_step(self._h)

    # ---- agents ----
    def _get2(self, i, what):
        import ctypes
        out = (ctypes.c_float * 2)()
        self._L.orc_sim_get_agent(self._h, int(i), what, out)
        return (out[0], out[1])

    def getNumAgents(self):
        return self._L.orc_sim_num_agents(self._h)

    def getAgentPosition(self, i):
        return self._get2(i, 0)

    def getAgentVelocity(self, i):
        return self._get2(i, 1)

    def getAgentPrefVelocity(self, i):
        return self._get2(i, 2)

    def setAgentPosition(self, i, p):
        self._L.orc_sim_set_agent(self._h, int(i), 0, p[0], p[1])

    def setAgentVelocity(self, i, v):
        self._L.orc_sim_set_agent(self._h, int(i), 1, v[0], v[1])

    def setAgentPrefVelocity(self, i, v):
        self._L.orc_sim_set_agent(self._h, int(i), 2, v[0], v[1])

    def getAgentNumAgentNeighbors(self, i):
        return self._L.orc_sim_num_agent_neighbors(self._h, int(i))

    def getAgentAgentNeighbor(self, i, k):
        return self._L.orc_sim_agent_neighbor(self._h, int(i), int(k))

    def getAgentNumObstacleNeighbors(self, i):
        return self._L.orc_sim_num_obstacle_neighbors(self._h, int(i))

    def getAgentObstacleNeighbor(self, i, k):
        return self._L.orc_sim_obstacle_neighbor(self._h, int(i), int(k))

    # ---- obstacles ----
    def getNextObstacleVertexNo(self, v):
        return self._L.orc_sim_next_obstacle_vertex(self._h, int(v))

    def getPrevObstacleVertexNo(self, v):
        return self._L.orc_sim_prev_obstacle_vertex(self._h, int(v))

    def getObstacleVertex(self, v):
        import ctypes
        out = (ctypes.c_float * 2)()
        self._L.orc_sim_obstacle_vertex(self._h, int(v), out)
        return (out[0], out[1])

    def getNumObstacleVertices(self):
        return self._L.orc_sim_num_obstacle_vertices(self._h)
