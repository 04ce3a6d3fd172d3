"""An independent fp64 restatement of the ORCA half-planes and of the velocity programme, from the GEOMETRY of velocity
obstacles (van den Berg, Guy, Lin, Manocha: "Reciprocal n-body collision avoidance", sections 4-5; the obstacle rules are
those of the RVO2 library the reference calls) -- written with angles, explicit tangent points and
point-to-feature distances, not with the closed forms and the operation order of App. A that the oracle and the kernels
follow.  CPU test infrastructure: tests/test_oracle_orca_definition.py compares the oracle with it value by value.

Conventions: a half-plane is (x0, n): the permitted velocities are { v : (v - x0) . n >= 0 }; n is a unit vector.  Every
function also returns a MARGIN: the distance (in the natural unit of the decision) to the nearest case switch -- fp32 and
fp64 may legitimately fall on different sides of a discontinuity, so scenes with a small margin are not compared.
"""
import math

import numpy as np

EPS = 1e-5          # RVO_EPSILON


def _rot(v, ang):
    c, s = math.cos(ang), math.sin(ang)
    return np.array([c * v[0] - s * v[1], s * v[0] + c * v[1]])


def _ang(v):
    return math.atan2(v[1], v[0])


def _wrap(a):
    return (a + math.pi) % (2 * math.pi) - math.pi


def _left(v):
    return np.array([-v[1], v[0]])


# ---------------------------------------------------------------------------------------------------------------------
# obstacles
# ---------------------------------------------------------------------------------------------------------------------
def obstacle_table(verts, nxt, n_original=None):
    """From the processed vertex table (positions + successor, as sim.getObstacleVertex / getNextObstacleVertexNo expose it):
    predecessor, unit edge directions and the convexity flag of every vertex -- derived here from the coordinates alone
    (a vertex is convex when the boundary turns left at it or goes straight on).  Vertices beyond the caller's own
    (n_original: the points where processObstacles cut an edge, appended behind them) lie ON an edge: straight on, convex by
    construction -- their rounded coordinates must not decide it."""
    verts = np.asarray(verts, np.float64)
    n = len(verts)
    prv = np.zeros(n, int)
    for i in range(n):
        prv[nxt[i]] = i
    unit = np.zeros((n, 2))
    for i in range(n):
        d = verts[nxt[i]] - verts[i]
        unit[i] = d / np.linalg.norm(d)
    convex = np.zeros(n, bool)
    for i in range(n):
        if nxt[nxt[i]] == i or (n_original is not None and i >= n_original):   # a free-standing wall: both ends convex; a cut point
            convex[i] = True
        else:
            a, b = verts[i] - verts[prv[i]], verts[nxt[i]] - verts[i]
            convex[i] = (a[0] * b[1] - a[1] * b[0]) >= 0.0
    return dict(verts=verts, next=np.asarray(nxt, int), prev=prv, unit=unit, convex=convex)


def _pt_seg(p, a, b):
    d = b - a
    t = float(np.dot(p - a, d) / np.dot(d, d))
    tc = min(1.0, max(0.0, t))
    return float(np.linalg.norm(p - (a + tc * d))), t


def obstacle_neighbours(tab, p, rng_obst):
    """Edges the agent is strictly to the right of whose supporting line and whose segment are both nearer than rng_obst,
    nearest segment first (equal distances -- two edges sharing their nearest vertex -- in edge order).  margin: to the range."""
    out, margin = [], np.inf
    V, nx = tab["verts"], tab["next"]
    for e in range(len(V)):
        a, b = V[e], V[nx[e]]
        u = tab["unit"][e]
        side = u[0] * (p[1] - a[1]) - u[1] * (p[0] - a[0])          # > 0: left of the edge
        dline = abs(side)
        dseg, _ = _pt_seg(p, a, b)
        if side < 0:
            margin = min(margin, abs(dseg - rng_obst), abs(dline - rng_obst) if dseg < rng_obst else np.inf)
        margin = min(margin, abs(side)) if dseg < rng_obst else margin
        if side < 0 and dline < rng_obst and dseg < rng_obst:
            out.append((dseg, e))
    out.sort(key=lambda t: (round(t[0], 9), t[1]))
    for k in range(len(out) - 1):     # two DIFFERENT distances closer than rounding could tell apart: the order is not defined
        gap = out[k + 1][0] - out[k][0]
        if 1e-9 < gap:
            margin = min(margin, gap * 10)
    return [e for _, e in out], margin


def _tangent_dir(c, r, side):
    """Unit direction of the ray from the origin tangent to the disc D(c, r), touching it on its left (side = +1) or right
    (side = -1) as seen from the origin."""
    d = np.linalg.norm(c)
    return _rot(c / d, side * math.asin(min(1.0, r / d)))


def obstacle_halfplane(tab, e, p, v, r, tau, earlier):
    """The half-plane edge e contributes for an agent at p with velocity v (or None), with the kind of boundary piece and the
    margin.  `earlier`: the half-planes already added for this agent (the already-covered rule).

    Geometry: VO = { x : the agent moving with x comes within r of the edge inside tau seconds } = the cone from the origin
    over the stadium (edge (+) disc(r)) / tau, cut off at the stadium's front.  Its skeleton in velocity space: the scaled
    edge [cL, cR] (cL = (o1 - p) / tau, ...) and the centre lines of the two legs (rays from cL / cR along the tangent
    directions); the boundary is the skeleton pushed out by r / tau.  The library's rule: take the skeleton feature nearest
    to v -- an end point (then the constraint is the tangent to the disc of radius r / tau round it, facing v), the inside of
    the scaled edge (the straight front) or the inside of a leg's centre ray (the leg) -- and make the tangent there the
    half-plane.  Polygon context: at a non-convex vertex the leg runs along the edge itself; a leg that the adjacent edge
    sticks out of is replaced by that edge's direction and yields no constraint of its own ("foreign": the adjacent edge
    supplies it); an edge seen end-on (the agent inside its thickness band, beyond an end) counts as its near end point only.
    """
    V, nx, pv, U, cvx = tab["verts"], tab["next"], tab["prev"], tab["unit"], tab["convex"]
    i1, i2 = e, nx[e]
    a, b = V[i1] - p, V[i2] - p                      # the end points relative to the agent
    u = U[e]
    rho = r / tau
    margin = np.inf
    # ---- already covered: both end discs of the scaled edge lie wholly on the forbidden side of an earlier half-plane ----
    for x0, n in earlier:
        da, db = -float(np.dot(a / tau - x0, n)), -float(np.dot(b / tau - x0, n))     # depth into the forbidden side
        # (the rule's own slack is EPS = 1e-5 and its typical case sits exactly on it: the next piece of a straight wall has
        # depth == rho.  The depths are O(1) values rounded to ~1e-7 in fp32, so a distance of 2e-6 from the threshold is
        # already thirty roundings: this margin is reported x 50 to meet the callers' common 1e-4 bar)
        if da >= rho - EPS and db >= rho - EPS:
            return None, "covered", 50.0 * min(da - (rho - EPS), db - (rho - EPS))
        margin = min(margin, 50.0 * max(abs(da - (rho - EPS)) if db >= rho - EPS - 2e-6 else np.inf,
                                        abs(db - (rho - EPS)) if da >= rho - EPS - 2e-6 else np.inf))
    dseg, s = _pt_seg(np.zeros(2), a, b)
    dline = abs(u[0] * a[1] - u[1] * a[0])
    da_, db_ = float(np.linalg.norm(a)), float(np.linalg.norm(b))
    margin = min(margin, abs(s), abs(s - 1.0))
    # ---- touching: the constraint is "do not move closer to the nearest point of the obstacle" ----
    if s < 0 and da_ <= r:
        margin = min(margin, r - da_)
        if not cvx[i1]:
            return None, "coll-vertex-nonconvex", margin
        return (np.zeros(2), -a / da_), "coll-vertex", margin
    if s > 1 and db_ <= r:
        margin = min(margin, r - db_)
        nxt_side = U[i2][0] * (-b[1]) - U[i2][1] * (-b[0])        # > 0: the agent is left of the edge leaving o2
        margin = min(margin, abs(nxt_side))
        if not (cvx[i2] and nxt_side >= 0):       # (if the agent sees that next edge from its open side, IT adds the line)
            return None, "coll-vertex-skipped", margin
        return (np.zeros(2), -b / db_), "coll-vertex", margin
    if 0 <= s < 1 and dline <= r:
        return (np.zeros(2), np.array([u[1], -u[0]])), "coll-segment", min(margin, r - dline)
    margin = min(margin, abs(dline - r), abs(da_ - r) if s < 0 else np.inf, abs(db_ - r) if s > 1 else np.inf)
    # ---- the skeleton ----
    left_foreign = right_foreign = False
    if s < 0 and dline <= r:                      # end-on, beyond o1
        if not cvx[i1]:
            return None, "oblique-nonconvex", margin
        i2 = i1; b = a
        L, Rg = _tangent_dir(a, r, +1), _tangent_dir(a, r, -1)
    elif s > 1 and dline <= r:                    # end-on, beyond o2
        if not cvx[i2]:
            return None, "oblique-nonconvex", margin
        i1 = i2; a = b
        L, Rg = _tangent_dir(b, r, +1), _tangent_dir(b, r, -1)
    else:
        L = _tangent_dir(a, r, +1) if cvx[i1] else -u
        Rg = _tangent_dir(b, r, -1) if cvx[i2] else u
    # a leg the adjacent edge sticks out of
    if cvx[i1]:
        back = -U[pv[i1]]                          # from o1 back along the edge that arrives there
        turn = _wrap(_ang(back) - _ang(L))         # >= 0: that edge lies to the left of the leg
        margin = min(margin, abs(math.sin(turn)))
        if math.sin(turn) >= 0:
            L, left_foreign = back, True
    if cvx[i2]:
        fwd = U[i2]
        turn = _wrap(_ang(fwd) - _ang(Rg))
        margin = min(margin, abs(math.sin(turn)))
        if math.sin(turn) <= 0:
            Rg, right_foreign = fwd, True
    cL, cR = a / tau, b / tau
    same = i1 == i2
    # nearest skeleton feature: distances to the scaled edge and to the two centre rays (each clamped at its start)
    tl, tr = float(np.dot(v - cL, L)), float(np.dot(v - cR, Rg))
    d_left = float(np.linalg.norm(v - (cL + max(tl, 0.0) * L)))
    d_right = float(np.linalg.norm(v - (cR + max(tr, 0.0) * Rg)))
    if same:
        t, d_cut = 0.5, np.inf
    else:
        d_cut, t = _pt_seg(v, cL, cR)
    margin = min(margin, abs(tl), abs(tr), (abs(t) if not same else np.inf), (abs(t - 1.0) if not same else np.inf))
    if (t < 0 and tl < 0) or (same and tl < 0 and tr < 0):
        w = (v - cL) / np.linalg.norm(v - cL)
        return (cL + rho * w, w), "circle-left", margin
    if t > 1 and tr < 0:
        w = (v - cR) / np.linalg.norm(v - cR)
        return (cR + rho * w, w), "circle-right", margin
    cands = []
    if not same and 0 <= t <= 1:
        cands.append((d_cut, 0, "cutoff"))
    if tl >= 0:
        cands.append((d_left, 1, "leg-left"))
    if tr >= 0:
        cands.append((d_right, 2, "leg-right"))
    cands.sort()
    if len(cands) > 1:
        margin = min(margin, cands[1][0] - cands[0][0])
    kind = cands[0][2]
    if kind == "cutoff":
        n = np.array([U[e][1], -U[e][0]])          # towards the agent's side of the edge
        return (cL + rho * n, n), kind, margin
    if kind == "leg-left":
        if left_foreign:
            return None, "leg-left-foreign", margin
        n = _left(L)
        return (cL + rho * n, n), kind, margin
    if right_foreign:
        return None, "leg-right-foreign", margin
    n = -_left(Rg)
    return (cR + rho * n, n), kind, margin


def obstacle_halfplanes(tab, p, v, r, tau, rng_obst):
    """[(edge, kind, halfplane or None)] in neighbour order, and the smallest margin."""
    order, margin = obstacle_neighbours(tab, p, rng_obst)
    out, planes = [], []
    for e in order:
        hp, kind, m = obstacle_halfplane(tab, e, p, v, r, tau, planes)
        margin = min(margin, m)
        out.append((e, kind, hp))
        if hp is not None:
            planes.append(hp)
    return out, margin


# ---------------------------------------------------------------------------------------------------------------------
# agents
# ---------------------------------------------------------------------------------------------------------------------
def agent_halfplane(pA, vA, pB, vB, rA, rB, tau, dt):
    """ORCA^tau_{A|B}: the boundary point of VO^tau closest to the relative velocity, u = the way there, the half-plane through
    vA + u / 2 with the outward normal.  Overlapping agents: the same with the horizon of ONE time step and the disc part of
    the boundary only (the library's rule: get apart within this step).  Returns (halfplane, kind, margin)."""
    rp, rv, cr = pB - pA, vA - vB, rA + rB
    d = float(np.linalg.norm(rp))
    if d <= cr:
        c, rho = rp / dt, cr / dt
        w = rv - c
        wl = float(np.linalg.norm(w))
        e = w / wl
        return (vA + 0.5 * ((c + rho * e) - rv), e), "collision", min(cr - d, wl)
    c, rho = rp / tau, cr / tau
    ax = rp / d
    alpha = math.asin(cr / d)
    cands = []
    t0 = float(np.linalg.norm(c)) * math.cos(alpha)          # distance of the tangent points from the apex
    for side in (+1.0, -1.0):
        leg = _rot(ax, side * alpha)
        s = float(np.dot(rv, leg))
        if s >= t0:
            cands.append((float(np.linalg.norm(rv - s * leg)), s * leg, side * _left(leg), "leg-left" if side > 0 else "leg-right", s - t0))
    w = rv - c
    wl = float(np.linalg.norm(w))
    if wl > 0:
        e = w / wl
        front = float(np.dot(e, -ax)) - math.sin(alpha)       # >= 0: on the arc that faces the origin
        if front >= 0:
            cands.append((abs(wl - rho), c + rho * e, e, "cutoff-circle", front))
    if not cands:
        return None, "none", 0.0
    cands.sort(key=lambda t: t[0])
    best = cands[0]
    margin = best[4]
    if len(cands) > 1:
        margin = min(margin, cands[1][0] - cands[0][0])
    if best[3] != "cutoff-circle":       # which leg: decided by the side of the axis the relative velocity is on
        margin = min(margin, abs(ax[0] * w[1] - ax[1] * w[0]))
    u = best[1] - rv
    return (vA + 0.5 * u, best[2]), best[3], margin


# ---------------------------------------------------------------------------------------------------------------------
# the programme
# ---------------------------------------------------------------------------------------------------------------------
def _disc_polygon(vmax, m=4096):
    """Outer polygon of the speed disc: m half-planes; every point of the disc satisfies them, a point that satisfies them is
    within vmax (1 / cos(pi / m) - 1) = 3e-7 vmax of it."""
    th = 2 * math.pi * np.arange(m) / m
    N = -np.stack([np.cos(th), np.sin(th)], 1)            # inward normals
    return N, -np.full(m, vmax)                            # n . v >= -vmax


def solve_feasible(planes, pref, vmax):
    """argmin |v - pref| over the half-planes and the disc (a strictly convex programme: unique), or None if empty: by
    scipy's SLSQP from several starts would be fragile -- instead the optimum of a 2-D projection problem is pref itself, its
    projection on one boundary, or a vertex of two boundaries: enumerate and take the best feasible candidate."""
    def ok(x, tol=1e-9):
        return x @ x <= vmax * vmax + tol and all((x - x0) @ n >= -tol for x0, n in planes)
    cands = [pref.copy()]
    npf = np.linalg.norm(pref)
    if npf > vmax:
        cands.append(pref / npf * vmax)
    lines = [(x0, np.array([n[1], -n[0]])) for x0, n in planes]
    for (x0, d) in lines:
        cands.append(x0 + ((pref - x0) @ d) * d)
        bq, cq = x0 @ d, x0 @ x0 - vmax * vmax
        disc = bq * bq - cq
        if disc >= 0:
            for sg in (-1.0, 1.0):
                cands.append(x0 + (-bq + sg * math.sqrt(disc)) * d)
    for i in range(len(lines)):
        for j in range(i + 1, len(lines)):
            (p1, d1), (p2, d2) = lines[i], lines[j]
            den = d1[0] * d2[1] - d1[1] * d2[0]
            if abs(den) > 1e-12:
                t = ((p2[0] - p1[0]) * d2[1] - (p2[1] - p1[1]) * d2[0]) / den
                cands.append(p1 + t * d1)
    good = [x for x in cands if ok(x)]
    if not good:
        return None
    return min(good, key=lambda x: (x - pref) @ (x - pref))


def solve_minimal_penetration(hard, soft, vmax, m=4096):
    """The library's fallback when the half-planes have no common point (paper, section 5.3; obstacle half-planes stay hard):
        minimise  z   subject to   -(v - x0_k) . n_k <= z  for the agent half-planes (soft),
                                    (v - x0_o) . n_o >= 0   for the obstacle half-planes (hard),   |v| <= vmax
    as a linear programme in (v, z) over the outer polygon of the disc (scipy.optimize.linprog, HiGHS).
    -> (z*, v*, spread): spread = how far apart near-optimal corners are (a flat optimum is not unique: compare values only)."""
    from scipy.optimize import linprog
    N, bnd = _disc_polygon(vmax, m)
    A, b = [], []
    for x0, n in soft:                       # -n . v - z <= -n . x0
        A.append([-n[0], -n[1], -1.0]); b.append(-float(n @ x0))
    for x0, n in hard:                       # -n . v <= -n . x0
        A.append([-n[0], -n[1], 0.0]); b.append(-float(n @ x0))
    A = np.concatenate([np.array(A).reshape(-1, 3), np.concatenate([-N, np.zeros((m, 1))], 1)])
    b = np.concatenate([np.array(b), -bnd])
    res = linprog([0.0, 0.0, 1.0], A_ub=A, b_ub=b, bounds=[(None, None)] * 3, method="highs")
    if res.status != 0:
        return None
    z, v = float(res.x[2]), np.array(res.x[:2])
    # uniqueness: the farthest point (along +-x, +-y) that is still within 1e-7 of the optimum
    spread = 0.0
    A2 = np.concatenate([A, [[0.0, 0.0, 1.0]]]); b2 = np.concatenate([b, [z + 1e-7]])
    for cvec in ([1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]):
        r2 = linprog(cvec, A_ub=A2, b_ub=b2, bounds=[(None, None)] * 3, method="highs")
        if r2.status == 0:
            spread = max(spread, float(np.linalg.norm(np.array(r2.x[:2]) - v)))
    return z, v, spread
