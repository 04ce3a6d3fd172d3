"""The oracle's ORCA arithmetic against the DEFINITIONS of the published algorithm (van den Berg, Guy, Lin, Manocha,
"Reciprocal n-body collision avoidance", ISRR 2009), restated here independently in numpy fp64 from the geometry -- not
from the operation order of SURVEY Appendix A that the oracle and the kernels follow:

  * VO^tau_{A|B} = { v : exists t in (0, tau] with |t v - (p_B - p_A)| < r_A + r_B }: a cone with its apex at the origin,
    truncated by the disc D((p_B - p_A) / tau, (r_A + r_B) / tau);
  * u = the vector from the relative velocity v_A - v_B to the closest point of the boundary of VO, n = the outward
    normal there;  ORCA^tau_{A|B} = { v : (v - (v_A + u / 2)) . n >= 0 };
  * the new velocity = the point of  (intersection of the ORCA half-planes)  n  D(0, v_max)  closest to the preferred
    velocity (a strictly convex programme in 2-D: its optimum is the preferred velocity itself, its projection onto one
    constraint boundary, or a vertex of two boundaries -- enumerated and checked here, no simplex order involved).

The reference delegates this to the absent third-party `rvo2` module, so the oracle cannot be pinned to its outputs
(parity unpinned for the ORCA arithmetic); this file pins neighbour half-planes (App. A.4) and the
feasible linear programme (App. A.5 LP1 / LP2) to the paper's geometry on thousands of random configurations.  Obstacle
half-planes (A.3) and the infeasible case (LP3) stay with the analytic cases of test_oracle_orca.py.
"""
import numpy as np
import pytest

from oracle.rvo2_shim import PyRVOSimulator

DT, TAU, R, VMAX = 1 / 60., 1.5, 0.5, 1.0


def vo_closest_boundary(rp, rv, cr, tau):
    """(closest boundary point of VO^tau to rv, outward unit normal there, margin to the nearest case switch).
    Geometry only: the boundary is the front arc of the disc D(c, rho), c = rp / tau, rho = cr / tau, between its two
    tangent points as seen from the origin, and the two tangent rays from those points away from the origin."""
    c, rho = rp / tau, cr / tau
    d = np.linalg.norm(rp)
    ax = rp / d                                   # cone axis
    perp = np.array([-ax[1], ax[0]])
    sin_a = cr / d                                # half-angle of the cone
    cos_a = np.sqrt(1.0 - sin_a * sin_a)
    cands = []
    # tangent points of the truncating disc and the leg directions (rays from the tangent points, away from the origin)
    for side in (+1.0, -1.0):
        leg_dir = cos_a * ax + side * sin_a * perp                 # unit vector along the leg
        t0 = np.linalg.norm(c) * cos_a                             # distance of the tangent point from the apex
        s = max(float(np.dot(rv, leg_dir)), t0)                    # projection, clamped to the ray
        pt = s * leg_dir
        # outward normal of the leg on `side`: perpendicular to leg_dir, pointing away from the cone's inside
        normal = side * np.array([-leg_dir[1], leg_dir[0]])
        cands.append((np.linalg.norm(rv - pt), pt, normal, "leg", s - t0))
    # front arc: points c + rho * e with e within the angular span facing the origin (between the tangent points)
    w = rv - c
    wl = np.linalg.norm(w)
    if wl > 0:
        e = w / wl
        # the arc spans directions e with  e . (-ax) >= sin_a  (the tangent points are at angle 90deg - alpha from -ax)
        if np.dot(e, -ax) >= sin_a:
            pt = c + rho * e
            cands.append((abs(wl - rho), pt, e, "arc", float(np.dot(e, -ax) - sin_a)))
    cands.sort(key=lambda t: t[0])
    best = cands[0]
    gap = cands[1][0] - cands[0][0] if len(cands) > 1 else np.inf    # how close the runner-up is (a case switch)
    return best[1], best[2], min(gap, abs(best[4]) if best[3] == "arc" else np.inf)


def orca_halfplane(pA, vA, pB, vB):
    rp, rv = pB - pA, vA - vB
    pt, n, margin = vo_closest_boundary(rp, rv, 2 * R, TAU)
    u = pt - rv
    return vA + 0.5 * u, n, margin          # feasible side: (v - point) . n >= 0


def solve_qp(planes, pref, vmax):
    """argmin |v - pref| over {(v - p_k) . n_k >= 0 for all k} n {|v| <= vmax}; None if the region is (numerically) empty.
    Candidates: pref, its projections onto each line and onto the circle, line-line and line-circle intersections."""
    def feasible(v, tol=1e-9):
        return np.dot(v, v) <= vmax * vmax + tol and all(np.dot(v - p, n) >= -tol for p, n in planes)
    cands = [pref.copy()]
    if np.linalg.norm(pref) > 0:
        cands.append(pref / np.linalg.norm(pref) * min(vmax, np.linalg.norm(pref)))
    lines = [(p, np.array([n[1], -n[0]])) for p, n in planes]          # point + direction
    for (p, dvec), (_, n) in zip(lines, planes):
        cands.append(p + np.dot(pref - p, dvec) * dvec)                  # projection onto the line
        b = np.dot(p, dvec); cc = np.dot(p, p) - vmax * vmax            # line-circle intersections
        disc = b * b - cc
        if disc >= 0:
            for sgn in (-1, 1):
                cands.append(p + (-b + sgn * np.sqrt(disc)) * dvec)
    for a in range(len(lines)):
        for b_ in range(a + 1, len(lines)):
            (p1, d1), (p2, d2) = lines[a], lines[b_]
            den = d1[0] * d2[1] - d1[1] * d2[0]
            if abs(den) > 1e-12:
                t = ((p2[0] - p1[0]) * d2[1] - (p2[1] - p1[1]) * d2[0]) / den
                cands.append(p1 + t * d1)
    good = [v for v in cands if feasible(v)]
    if not good:
        return None
    return min(good, key=lambda v: np.dot(v - pref, v - pref))


def random_scene(rng, n):
    while True:
        pos = rng.uniform(0, 3.5 + 0.6 * n, (n, 2))
        dd = np.linalg.norm(pos[:, None] - pos[None], axis=2) + np.eye(n) * 10
        if dd.min() > 2 * R + 0.05:
            break
    ang, spd = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 1, n)
    vel = np.stack([np.cos(ang), np.sin(ang)], 1) * spd[:, None]
    ang, spd = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 1.3, n)
    pref = np.stack([np.cos(ang), np.sin(ang)], 1) * spd[:, None]
    return pos.astype(np.float32), vel.astype(np.float32), pref.astype(np.float32)


@pytest.mark.parametrize("n_agents,n_scenes", [(2, 1500), (3, 600), (5, 300), (8, 120)])
def test_new_velocity_is_the_papers_optimum(n_agents, n_scenes):
    rng = np.random.RandomState(100 + n_agents)
    checked = skipped_switch = infeasible = 0
    worst = 0.0
    for _ in range(n_scenes):
        pos, vel, pref = random_scene(rng, n_agents)
        s = PyRVOSimulator(timeStep=DT, neighborDist=100.0, maxNeighbors=n_agents - 1, timeHorizon=TAU,
                           timeHorizonObst=TAU, radius=R, maxSpeed=VMAX)
        for i in range(n_agents):
            s.addAgent((float(pos[i, 0]), float(pos[i, 1])))
            s.setAgentVelocity(i, (float(vel[i, 0]), float(vel[i, 1])))
            s.setAgentPrefVelocity(i, (float(pref[i, 0]), float(pref[i, 1])))
        s.doStep()
        p64, v64, f64 = pos.astype(np.float64), vel.astype(np.float64), pref.astype(np.float64)
        for i in range(n_agents):
            planes, margin = [], np.inf
            for j in range(n_agents):
                if j != i:
                    pt, n, m = orca_halfplane(p64[i], v64[i], p64[j], v64[j])
                    planes.append((pt, n)); margin = min(margin, m)
            if margin < 1e-3:           # the relative velocity sits on a switch between arc and leg / left and right:
                skipped_switch += 1     # fp32 and fp64 may legitimately take different sides of the discontinuity
                continue
            ref = solve_qp(planes, f64[i], VMAX)
            if ref is None:             # infeasible: LP3 territory, not this test's subject
                infeasible += 1
                continue
            got = np.array(s.getAgentVelocity(i), np.float64)
            err = np.linalg.norm(got - ref)
            worst = max(worst, err)
            assert err < 5e-6, (i, pos, vel, pref, got, ref)   # observed worst: 6e-7 (fp32 rounding)
            checked += 1
    assert checked > 0.7 * n_agents * n_scenes, (checked, skipped_switch, infeasible)


def test_halfplane_matches_geometry_in_the_three_regimes():
    """One configuration per regime of the boundary (front arc, left leg, right leg), half-plane from the geometry against
    the oracle's result for a preferred velocity that violates it (the result then lies ON the half-plane's line)."""
    cases = [  # pB, vA, vB: chosen so that rv - c points back at the origin (arc), or rv is inside the cone left / right of the axis
        ((3.0, 0.0), (0.1, 0.0), (-0.1, 0.0)),      # slow approach: closest boundary point on the front arc
        ((2.0, 0.0), (1.0, 0.3), (-0.6, 0.0)),      # fast, left of the axis: left leg
        ((2.0, 0.0), (1.0, -0.3), (-0.6, 0.0)),     # fast, right of the axis: right leg
    ]
    kinds = []
    for pB, vA, vB in cases:
        pA = np.zeros(2); pB = np.array(pB); vA = np.array(vA); vB = np.array(vB)
        pt, n, margin = orca_halfplane(pA, vA, pB, vB)
        assert margin > 1e-3
        rp, rv = pB - pA, vA - vB
        c = rp / TAU
        kinds.append("arc" if abs(np.linalg.norm((rv + 2 * (pt - vA)) - c) - 2 * R / TAU) < 1e-9 else "leg")
        pref = pt - 0.2 * n + 0.05 * np.array([n[1], -n[0]])       # on the forbidden side of the half-plane
        ref = solve_qp([(pt, n)], pref, VMAX)
        s = PyRVOSimulator(timeStep=DT, neighborDist=100.0, maxNeighbors=1, timeHorizon=TAU, timeHorizonObst=TAU,
                           radius=R, maxSpeed=VMAX)
        s.addAgent((0.0, 0.0)); s.addAgent((float(pB[0]), float(pB[1])))
        s.setAgentVelocity(0, tuple(map(float, vA))); s.setAgentVelocity(1, tuple(map(float, vB)))
        s.setAgentPrefVelocity(0, tuple(map(float, pref))); s.setAgentPrefVelocity(1, tuple(map(float, vB)))
        s.doStep()
        got = np.array(s.getAgentVelocity(0), np.float64)
        assert abs(np.dot(got - pt, n)) < 2e-6, (got, pt, n)        # the half-plane is active: the result lies on its line
        assert np.linalg.norm(got - ref) < 5e-6, (got, ref)
    assert kinds == ["arc", "leg", "leg"]
