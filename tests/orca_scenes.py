"""Single-step ORCA scenes that between them take every branch of SURVEY App. A.3 (obstacle half-planes), A.4 (agent
half-planes) and A.5 (LP1 / LP2 / LP3): the differential scenes of tests/test_oracle_orca_definition.py (oracle against an
independent fp64 restatement) and of tests/test_gpu_orca_scenes.py (HIP kernels against the oracle on the very same scenes).

A scene = one arena at one instant: polygons (RVO2 convention: counter-clockwise = an obstacle seen from outside, clockwise
= a boundary seen from inside, two vertices = a free-standing wall), agent positions / velocities / preferred velocities
(fp32), and `focus`: the agent whose half-planes and new velocity are examined (the others only matter as its neighbours).
Everything is drawn from seeded generators, so CPU and GPU suites see identical inputs; units are the reference's (metres,
seconds: radius 0.5, maxSpeed 1, horizons 1.5 s, dt 1/60 -- collision_avoidence_env.py:27-32, 130).

Families (what each is built to provoke):
  walls     free-standing walls seen from every side: broadside, end-on ("oblique"), past an end, touching (collision branches)
  convex    convex obstacles seen across a corner: two visible faces -> foreign legs, projections on them (skipped lines)
  notch     an L-shaped obstacle: a non-convex vertex, edges cut by processObstacles, legs along the edge
  room      clockwise boundaries (every vertex non-convex from inside), subdivided walls (already-covered edges), agents
            outside near a corner (the non-convex collision / oblique skips), corridors (anti-parallel lines in LP1)
  crowd     agents only: cut-off circle / left leg / right leg, overlapping pairs (the 1 / dt branch), infeasible rings (LP3)
  mirror    exactly mirrored neighbours on one axis: parallel and anti-parallel agent lines (the |det| <= eps branches of LP1 / LP3)
  hemmed    agents pressed against walls by a crowd: LP3 with obstacle lines as hard constraints
"""
import math

import numpy as np

R, VMAX, TAU, TAU_OBST, DT = 0.5, 1.0, 1.5, 1.5, 1 / 60.
NEIGHBOR_DIST = 50.0
RANGE_OBST = TAU_OBST * VMAX + R


def _f32(a):
    return np.asarray(a, np.float32)


def _unit(ang):
    return np.array([math.cos(ang), math.sin(ang)])


def _vel(rng, lo=0.0, hi=1.2):
    return rng.uniform(lo, hi) * _unit(rng.uniform(0, 2 * math.pi))


def _scene(family, polys, pos, vel, pref, focus=0):
    pos, vel, pref = _f32(pos).reshape(-1, 2), _f32(vel).reshape(-1, 2), _f32(pref).reshape(-1, 2)
    assert pos.shape == vel.shape == pref.shape
    return dict(family=family, polys=[_f32(q).reshape(-1, 2) for q in polys], pos=pos, vel=vel, pref=pref, focus=focus)


def walls(rng, n):
    out = []
    for k in range(n):
        half = rng.uniform(0.4, 2.5)
        ang = rng.uniform(0, 2 * math.pi)
        u = _unit(ang)
        nrm = np.array([-u[1], u[0]])
        c = rng.uniform(3, 7, 2)
        a, b = c - half * u, c + half * u
        mode = k % 6
        if mode == 0:      # broadside
            p = c + rng.uniform(-half, half) * u + rng.choice([-1, 1]) * rng.uniform(0.55, 1.9) * nrm
        elif mode == 1:    # end-on: within the wall's thickness band, beyond an end
            end, sgn = (a, -1) if rng.rand() < 0.5 else (b, 1)
            p = end + sgn * rng.uniform(0.55, 1.8) * u + rng.uniform(-0.45, 0.45) * nrm
        elif mode == 2:    # past an end, off the band
            end, sgn = (a, -1) if rng.rand() < 0.5 else (b, 1)
            p = end + sgn * rng.uniform(0.0, 1.2) * u + rng.choice([-1, 1]) * rng.uniform(0.55, 1.5) * nrm
        elif mode == 3:    # touching the wall's face
            p = c + rng.uniform(-half, half) * u + rng.choice([-1, 1]) * rng.uniform(0.05, 0.48) * nrm
        elif mode == 4:    # touching an end
            end, sgn = (a, -1) if rng.rand() < 0.5 else (b, 1)
            p = end + rng.uniform(0.05, 0.48) * _unit(ang + (0 if sgn > 0 else math.pi) + rng.uniform(-1.5, 1.5))
        else:              # anywhere in range
            p = c + rng.uniform(-half - 1.5, half + 1.5) * u + rng.uniform(-1.9, 1.9) * nrm
        v = _vel(rng)
        if k % 3 == 0:     # heading for the wall: the current velocity inside the velocity obstacle
            tgt = c + rng.uniform(-half, half) * u - p
            v = rng.uniform(0.3, 1.1) * tgt / max(1e-9, np.linalg.norm(tgt)) + rng.uniform(-0.2, 0.2, 2)
        out.append(_scene("walls", [[a, b]], [p], [v], [_vel(rng, 0, 1.3)]))
    return out


def _convex_poly(rng):
    while True:
        m = rng.randint(3, 7)
        ang = np.sort(rng.uniform(0, 2 * math.pi, m))
        if np.min(np.diff(np.concatenate([ang, [ang[0] + 2 * math.pi]]))) > 0.55:
            break
    rad = rng.uniform(0.8, 2.2)
    c = rng.uniform(4, 8, 2)
    return c + rad * np.stack([np.cos(ang), np.sin(ang)], 1), c, rad      # counter-clockwise


def convex(rng, n):
    out = []
    for k in range(n):
        poly, c, rad = _convex_poly(rng)
        m = len(poly)
        j = rng.randint(m)
        corner = poly[j]
        outward = (corner - c) / np.linalg.norm(corner - c)
        if k % 4 == 3:     # touching a corner / a face
            p = corner + rng.uniform(0.1, 0.48) * _unit(math.atan2(outward[1], outward[0]) + rng.uniform(-1.2, 1.2))
        elif k % 4 == 2:   # in front of a face
            a, b = poly[j], poly[(j + 1) % m]
            e = (b - a) / np.linalg.norm(b - a)
            p = a + rng.uniform(0.1, 0.9) * (b - a) + rng.uniform(0.55, 1.8) * np.array([e[1], -e[0]])
        else:              # across a corner: two faces visible
            p = corner + rng.uniform(0.6, 1.9) * _unit(math.atan2(outward[1], outward[0]) + rng.uniform(-0.9, 0.9))
        v = _vel(rng)
        if k % 2 == 0:
            tgt = c + rng.uniform(-0.5, 0.5, 2) * rad - p
            v = rng.uniform(0.3, 1.1) * tgt / np.linalg.norm(tgt) + rng.uniform(-0.25, 0.25, 2)
        out.append(_scene("convex", [poly], [p], [v], [_vel(rng, 0, 1.3)]))
    return out


def notch(rng, n):
    out = []
    for k in range(n):
        s = rng.uniform(1.2, 2.4)
        o = rng.uniform(3, 6, 2)
        rot = rng.uniform(0, 2 * math.pi)
        cs, sn = math.cos(rot), math.sin(rot)
        L = np.array([[0, 0], [2 * s, 0], [2 * s, s], [s, s], [s, 2 * s], [0, 2 * s]], np.float64)       # ccw, notch at (s, s)
        poly = o + L @ np.array([[cs, sn], [-sn, cs]])
        nv = o + np.array([s, s]) @ np.array([[cs, sn], [-sn, cs]])
        diag = _unit(rot + math.pi / 4)
        if k % 3 == 0:     # deep in the notch
            p = nv + rng.uniform(0.72, 1.6) * diag + rng.uniform(-0.15, 0.15, 2)
        elif k % 3 == 1:   # along one of the notch's walls
            w = _unit(rot) if rng.rand() < 0.5 else _unit(rot + math.pi / 2)
            p = nv + rng.uniform(0.2, 0.95 * s) * w + rng.uniform(0.55, 1.6) * (diag * math.sqrt(2) - w)
        else:              # anywhere around
            p = o + (np.array([s, s]) + rng.uniform(-2.2 * s, 2.2 * s, 2)) @ np.array([[cs, sn], [-sn, cs]])
        out.append(_scene("notch", [poly], [p], [_vel(rng)], [_vel(rng, 0, 1.3)]))
    return out


def _subdivide(poly, parts):
    out = []
    m = len(poly)
    for k in range(m):
        a, b = poly[k], poly[(k + 1) % m]
        for t in range(parts):
            out.append(a + (b - a) * (t / parts))
    return np.array(out)


def room(rng, n):
    out = []
    for k in range(n):
        mode = k % 6
        w, h = rng.uniform(3, 6), rng.uniform(3, 6)
        if mode == 3:      # corridor: two walls in range at once
            w, h = rng.uniform(1.3, 3.2), rng.uniform(5, 8)
        o = rng.uniform(1, 3, 2)
        box = o + np.array([[0, 0], [0, h], [w, h], [w, 0]], np.float64)              # clockwise: seen from inside
        poly = _subdivide(box, rng.randint(2, 5)) if mode in (1, 4) else box
        if mode in (0, 1):     # inside, near a wall / a corner
            p = o + np.array([rng.uniform(0.55, min(w - 0.55, 2.0)), rng.uniform(0.55, min(h - 0.55, 2.0))])
            if rng.rand() < 0.5:
                p = o + np.array([w, h]) - (p - o)
        elif mode == 2:        # OUTSIDE, touching a corner: the non-convex vertex collision rules
            corner = box[rng.randint(4)]
            p = corner + rng.uniform(0.05, 0.48) * _unit(rng.uniform(0, 2 * math.pi))
        elif mode == 5:        # OUTSIDE, in line with a wall beyond its end: the non-convex oblique rules
            j = rng.randint(4)
            corner, other = box[j], box[(j + (1 if rng.rand() < 0.5 else 3)) % 4]
            u = (corner - other) / np.linalg.norm(corner - other)
            p = corner + rng.uniform(0.55, 1.6) * u + rng.uniform(-0.45, 0.45) * np.array([-u[1], u[0]])
        elif mode == 3:
            p = o + np.array([rng.uniform(0.52, w - 0.52), rng.uniform(1.0, h - 1.0)])
        else:                  # touching a wall from inside
            p = o + np.array([rng.uniform(0.05, 0.45), rng.uniform(0.6, h - 0.6)])
        v = _vel(rng)
        if k % 2:
            v = rng.uniform(0.4, 1.1) * _unit(rng.uniform(0, 2 * math.pi))
        out.append(_scene("room", [poly], [p], [v], [_vel(rng, 0.3, 1.3)]))
    return out


def _separated(rng, n, lo, hi, gap):
    while True:
        pos = rng.uniform(lo, hi, (n, 2))
        dd = np.linalg.norm(pos[:, None] - pos[None], axis=2) + np.eye(n) * 1e3
        if dd.min() > gap:
            return pos


def crowd(rng, n):
    out = []
    for k in range(n):
        mode = k % 4
        m = rng.randint(2, 9)
        if mode == 0:       # loose: feasible programmes, every boundary regime
            pos = _separated(rng, m, 2, 5.5 + 0.6 * m, 2 * R + 0.05)
            vel = np.array([_vel(rng, 0, 1.0) for _ in range(m)])
        elif mode == 1:     # a ring closing in on agent 0: infeasible (LP3)
            ang = np.sort(rng.uniform(0, 2 * math.pi, m - 1)) if m > 2 else np.array([rng.uniform(0, 6.28)])
            rad = rng.uniform(1.08, 2.0, m - 1)
            while True:
                pos = np.concatenate([[[6.0, 6.0]], 6.0 + np.stack([np.cos(ang), np.sin(ang)], 1) * rad[:, None]])
                dd = np.linalg.norm(pos[:, None] - pos[None], axis=2) + np.eye(m) * 1e3
                if dd.min() > 2 * R + 0.03:
                    break
                ang = np.sort(rng.uniform(0, 2 * math.pi, m - 1)); rad = rng.uniform(1.08, 2.2, m - 1)
            vel = np.concatenate([[rng.uniform(-0.3, 0.3, 2)],
                                  -(pos[1:] - pos[0]) / rad[:, None] * rng.uniform(0.6, 1.0, (m - 1, 1)) + rng.uniform(-0.15, 0.15, (m - 1, 2))])
        elif mode == 2:     # overlapping starts: the 1 / dt branch, far-away lines, LP3 that cannot improve
            pos = 6.0 + rng.uniform(-0.9, 0.9, (m, 2))
            vel = np.array([_vel(rng, 0, 1.0) for _ in range(m)])
        else:               # dense but separate
            pos = _separated(rng, m, 4, 5.2 + 0.45 * m, 2 * R + 0.02)
            vel = np.array([_vel(rng, 0.3, 1.0) for _ in range(m)])
        pref = np.array([_vel(rng, 0, 1.3) for _ in range(m)])
        out.append(_scene("crowd", [], pos, vel, pref))
    return out


def mirror(rng, n):
    """Neighbours of agent 0 on ONE axis through it, with velocities along that axis: their half-planes are exactly parallel
    or anti-parallel in fp32 (every y component is an exact zero)."""
    out = []
    for k in range(n):
        d1, d2 = rng.uniform(1.1, 2.4), rng.uniform(1.1, 2.4)
        s1, s2 = rng.uniform(0.0, 1.0), rng.uniform(0.0, 1.0)
        x0 = 6.0
        if k % 4 == 0:      # one on each side, closing in: disjoint anti-parallel half-planes
            pos = [[x0, 5.0], [x0 + d1, 5.0], [x0 - d2, 5.0]]
            vel = [[0.0, 0.0], [-s1, 0.0], [s2, 0.0]]
        elif k % 4 == 1:    # two on the same side: parallel half-planes, same direction
            pos = [[x0, 5.0], [x0 + d1, 5.0], [x0 + d1 + 1.05 + d2, 5.0]]
            vel = [[rng.uniform(0.0, 0.6), 0.0], [-s1, 0.0], [-s2, 0.0]]
        elif k % 4 == 2:    # both: one each side and a second one behind the first
            pos = [[x0, 5.0], [x0 + d1, 5.0], [x0 - d2, 5.0], [x0 + d1 + 1.05 + d2, 5.0]]
            vel = [[0.0, 0.0], [-s1, 0.0], [s2, 0.0], [-1.0, 0.0]]
        else:               # a slow near one and a fast far one on the same side, a closing one on the other: in LP3 the far one's
            #                 line meets an earlier line of its own direction (skipped) and one of the opposite direction
            d1, gap = rng.uniform(1.1, 1.5), rng.uniform(1.02, 1.2)
            pos = [[x0, 5.0], [x0 + d1, 5.0], [x0 - rng.uniform(d1 + 0.05, 1.9), 5.0], [x0 + d1 + gap, 5.0]]
            vel = [[0.0, 0.0], [rng.uniform(-0.1, 0.1), 0.0], [rng.uniform(0.7, 1.0), 0.0], [-rng.uniform(0.8, 1.0), 0.0]]
        m = len(pos)
        pref = np.array([_vel(rng, 0, 1.2) for _ in range(m)])
        if k % 2:
            pref[0] = [rng.uniform(-1, 1), 0.0]
        out.append(_scene("mirror", [], pos, vel, pref))
    return out


def hemmed(rng, n):
    out = []
    for k in range(n):
        w, h = rng.uniform(4, 7), rng.uniform(4, 7)
        o = rng.uniform(1, 2, 2)
        box = o + np.array([[0, 0], [0, h], [w, h], [w, 0]], np.float64)
        corner_mode = k % 2
        p0 = o + (np.array([rng.uniform(0.55, 1.1), rng.uniform(0.55, 1.1)]) if corner_mode else
                  np.array([rng.uniform(0.55, 1.0), rng.uniform(1.8, h - 1.8)]))
        m = rng.randint(3, 7)
        lo, hi = (0.05, math.pi / 2 - 0.05) if corner_mode else (-math.pi / 2 + 0.1, math.pi / 2 - 0.1)
        for _ in range(200):
            ang = np.sort(rng.uniform(lo, hi, m - 1))
            rad = rng.uniform(1.05, 1.9, m - 1)
            pos = np.concatenate([[p0], p0 + np.stack([np.cos(ang), np.sin(ang)], 1) * rad[:, None]])
            dd = np.linalg.norm(pos[:, None] - pos[None], axis=2) + np.eye(m) * 1e3
            inside = (pos[:, 0] > o[0] + 0.5).all() and (pos[:, 1] > o[1] + 0.5).all() and (pos[:, 0] < o[0] + w - 0.5).all() and \
                (pos[:, 1] < o[1] + h - 0.5).all()
            if dd.min() > 2 * R + 0.03 and inside:
                break
        else:
            continue
        vel = np.concatenate([[rng.uniform(-0.4, 0.4, 2)],
                              -(pos[1:] - pos[0]) / rad[:, None] * rng.uniform(0.7, 1.0, (m - 1, 1)) + rng.uniform(-0.1, 0.1, (m - 1, 2))])
        pref = np.array([_vel(rng, 0.2, 1.2) for _ in range(m)])
        out.append(_scene("hemmed", [box], pos, vel, pref))
    return out


FAMILIES = (("walls", walls, 1500, 11), ("convex", convex, 1200, 12), ("notch", notch, 900, 13), ("room", room, 3000, 14),
            ("crowd", crowd, 1600, 15), ("mirror", mirror, 800, 16), ("hemmed", hemmed, 1200, 17))


def all_scenes(scale=1.0):
    """Every family, seeded: the same list in every process."""
    out = []
    for name, fn, n, seed in FAMILIES:
        out += fn(np.random.RandomState(seed), max(6, int(n * scale)))
    return out


def run_oracle_sim(scene, capture=True):
    """One doStep of the scene through the oracle's simulator (oracle/rvo2_shim.py); returns (new velocity of the focus agent,
    its captured lines) -- CPU test infrastructure."""
    from oracle import oracle as o
    from oracle.rvo2_shim import PyRVOSimulator
    n = len(scene["pos"])
    s = PyRVOSimulator(timeStep=DT, neighborDist=NEIGHBOR_DIST, maxNeighbors=max(1, n - 1), timeHorizon=TAU,
                       timeHorizonObst=TAU_OBST, radius=R, maxSpeed=VMAX)
    for i in range(n):
        s.addAgent((float(scene["pos"][i, 0]), float(scene["pos"][i, 1])))
        s.setAgentVelocity(i, (float(scene["vel"][i, 0]), float(scene["vel"][i, 1])))
        s.setAgentPrefVelocity(i, (float(scene["pref"][i, 0]), float(scene["pref"][i, 1])))
    for q in scene["polys"]:
        s.addObstacle([tuple(map(float, v)) for v in q])
    if scene["polys"]:
        s.processObstacles()
    if capture:
        o.capture_next(0, scene["focus"])
    s.doStep()
    cap = o.captured() if capture else None
    if capture:
        o.capture_next(-1, -1)
    return np.array(s.getAgentVelocity(scene["focus"]), np.float64), cap, s
