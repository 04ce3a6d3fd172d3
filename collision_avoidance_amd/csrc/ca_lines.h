// ca_lines.h -- ORCA half-planes of agent neighbours (App. A.4) and obstacle edges (App. A.3)
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
#pragma once
#include "ca_common.h"

namespace ca {

// App. A.4: the half-plane induced by one neighbouring agent (both agents have radius R).
// Branch-free: a wave of 64 agents takes all three cases of the contract (cut-off circle, legs, collision) for nearly
// every neighbour slot anyway, and as branches they cost 179 vector instructions per line, 39 of them register copies
// at the joins.  Here the cut-off circle and the collision case share their arithmetic (the same formulas on the time
// scale 1/timeHorizon or 1/timeStep), the two legs share theirs, and selects pick the result.  Every value that is
// used is produced by the contract's own operations in the contract's order:
//   * right leg:  -( (rp.x leg + rp.y cr, -rp.x cr + rp.y leg) / d2 )  is evaluated as  -( (rp.x leg - rp.y c, rp.x c +
//     rp.y leg) / d2 ) with c = -cr: x - (-y) = x + y and (-x) y = x (-y) are exact identities of IEEE arithmetic;
//   * lanes on the other side of a select compute NaN / inf at worst (sqrt of a negative, 1/0), which is dropped.
__device__ __forceinline__ Line agent_orca_line(V2 pos, V2 vel, V2 opos, V2 ovel, float R, float invT, float invDt) {
    const V2 rp = opos - pos;
    const V2 rv = vel - ovel;
    const float distSq = absSq(rp);
    const float cr = R + R;
    const float crSq = sqr(cr);
    const bool coll = !(distSq > crSq);
    const float invX = coll ? invDt : invT;
    const V2 w = rv - invX * rp;
    const float wLenSq = absSq(w);
    const float dp1 = dot(w, rp);
    const bool circle = coll || (dp1 < 0.0f && sqr(dp1) > crSq * wLenSq);
    // cut-off circle / collision
    const float wLen = sqrt_ir(wLenSq);   // (0 or >= 1e-16: squares of differences of O(1) fp32 values)
    const V2 unitW = vdiv_ir(w, wLen);   // (wLen: 0 -- NaN either way, as the contract's 0 / 0 -- or >= 1e-8: differences of O(1) fp32 values)
    const V2 uC = (cr * invX - wLen) * unitW;
    // legs
    const float leg = sqrt_ir(distSq - crSq);   // (used where positive: >= an ulp of 1; negative -> NaN, unused)
    const bool left = det(rp, w) > 0.0f;
    const float c = left ? cr : -cr;
    V2 dirL = vdiv_ir(mk(rp.x * leg - rp.y * c, rp.x * c + rp.y * leg), distSq);   // (used for distSq > (2 R)^2 only)
    dirL = left ? dirL : -dirL;
    const float dp2 = dot(rv, dirL);
    const V2 uL = dp2 * dirL - rv;
    Line line;
    line.dir = circle ? mk(unitW.y, -unitW.x) : dirL;
    const V2 u = circle ? uC : uL;
    line.point = vel + 0.5f * u;
    return line;
}

// App. A.3: the half-plane induced by the obstacle edge e.  Returns false when the edge yields no
// line (already covered, non-convex vertex, foreign leg).  "o1"/"o2" are the edge's two vertices;
// the oblique cases collapse the edge onto one of them, exactly as the contract's o2<-o1 / o1<-o2.
// `covered(a, b)`: true when some earlier obstacle line already excludes both scaled end points.
template <class CoveredFn>
__device__ __forceinline__ bool obst_orca_line4(const ObstDev* __restrict__ tab, int e, V2 pos, V2 vel, float R,
                                                float invTO, CoveredFn covered, float& lpx, float& lpy, float& ldx, float& ldy) {
    // (the line leaves through four scalars: as a struct written on a dozen return paths it stayed in scratch memory)
#define CA_SET_LINE(P, D) do { const V2 _p = (P); const V2 _d = (D); lpx = _p.x; lpy = _p.y; ldx = _d.x; ldy = _d.y; } while (0)
    const ObstDev E = load_obst(tab, e);
    V2 o1p = mk(E.px, E.py), o2p = mk(E.qx, E.qy);
    V2 o1u = mk(E.ux, E.uy), o2u = mk(E.qux, E.quy);
    V2 lnu = mk(E.pux, E.puy);  // unitDir of o1's left neighbour (its prev vertex)
    bool o1c = E.convex != 0, o2c = E.qconvex != 0;
    bool same = false;
    const V2 rp1 = o1p - pos;
    const V2 rp2 = o2p - pos;
    if (covered(invTO * rp1, invTO * rp2)) return false;
    const float distSq1 = absSq(rp1), distSq2 = absSq(rp2), radiusSq = sqr(R);
    const V2 ov = o2p - o1p;
    const float s = div_ir(dot(-rp1, ov), absSq(ov));   // (an edge has a length)
    const float distSqLine = absSq(-rp1 - s * ov);
    if (s < 0.0f && distSq1 <= radiusSq) {
        if (o1c) {
            CA_SET_LINE(mk(0.0f, 0.0f), normalize(mk(-rp1.y, rp1.x)));
            return true;
        }
        return false;
    } else if (s > 1.0f && distSq2 <= radiusSq) {
        if (o2c && det(rp2, o2u) >= 0.0f) {
            CA_SET_LINE(mk(0.0f, 0.0f), normalize(mk(-rp2.y, rp2.x)));
            return true;
        }
        return false;
    } else if (s >= 0.0f && s < 1.0f && distSqLine <= radiusSq) {
        CA_SET_LINE(mk(0.0f, 0.0f), -o1u);
        return true;
    }
    V2 leftLeg, rightLeg;
    if (s < 0.0f && distSqLine <= radiusSq) {
        if (!o1c) return false;
        o2p = o1p; o2u = o1u; o2c = o1c; same = true;  // o2 <- o1
        const float leg1 = sqrt_ir(distSq1 - radiusSq);   // (no collision with the vertex in this branch: positive)
        leftLeg = vdiv_ir(mk(rp1.x * leg1 - rp1.y * R, rp1.x * R + rp1.y * leg1), distSq1);
        rightLeg = vdiv_ir(mk(rp1.x * leg1 + rp1.y * R, -rp1.x * R + rp1.y * leg1), distSq1);
    } else if (s > 1.0f && distSqLine <= radiusSq) {
        if (!o2c) return false;
        lnu = o1u;                                     // the new o1's prev vertex is the old o1
        o1p = o2p; o1u = o2u; o1c = o2c; same = true;  // o1 <- o2
        const float leg2 = sqrt_ir(distSq2 - radiusSq);
        leftLeg = vdiv_ir(mk(rp2.x * leg2 - rp2.y * R, rp2.x * R + rp2.y * leg2), distSq2);
        rightLeg = vdiv_ir(mk(rp2.x * leg2 + rp2.y * R, -rp2.x * R + rp2.y * leg2), distSq2);
    } else {
        if (o1c) {
            const float leg1 = sqrt_ir(distSq1 - radiusSq);
            leftLeg = vdiv_ir(mk(rp1.x * leg1 - rp1.y * R, rp1.x * R + rp1.y * leg1), distSq1);
        } else {
            leftLeg = -o1u;
        }
        if (o2c) {
            const float leg2 = sqrt_ir(distSq2 - radiusSq);
            rightLeg = vdiv_ir(mk(rp2.x * leg2 + rp2.y * R, -rp2.x * R + rp2.y * leg2), distSq2);
        } else {
            rightLeg = o1u;
        }
    }
    bool leftForeign = false, rightForeign = false;
    if (o1c && det(leftLeg, -lnu) >= 0.0f) {
        leftLeg = -lnu;
        leftForeign = true;
    }
    if (o2c && det(rightLeg, o2u) <= 0.0f) {
        rightLeg = o2u;
        rightForeign = true;
    }
    const V2 leftCut = invTO * (o1p - pos);
    const V2 rightCut = invTO * (o2p - pos);
    const V2 cutVec = rightCut - leftCut;
    const float t = same ? 0.5f : div_ir(dot(vel - leftCut, cutVec), absSq(cutVec));   // (not `same`: the cut-off segment has a length)
    const float tLeft = dot(vel - leftCut, leftLeg);
    const float tRight = dot(vel - rightCut, rightLeg);
    if ((t < 0.0f && tLeft < 0.0f) || (same && tLeft < 0.0f && tRight < 0.0f)) {
        const V2 unitW = normalize_ir(vel - leftCut);
        CA_SET_LINE(leftCut + R * invTO * unitW, mk(unitW.y, -unitW.x));
        return true;
    } else if (t > 1.0f && tRight < 0.0f) {
        const V2 unitW = normalize_ir(vel - rightCut);
        CA_SET_LINE(rightCut + R * invTO * unitW, mk(unitW.y, -unitW.x));
        return true;
    }
    const float INF = __int_as_float(0x7f800000);
    const float dCut = (t < 0.0f || t > 1.0f || same) ? INF : absSq(vel - (leftCut + t * cutVec));
    const float dLeft = (tLeft < 0.0f) ? INF : absSq(vel - (leftCut + tLeft * leftLeg));
    const float dRight = (tRight < 0.0f) ? INF : absSq(vel - (rightCut + tRight * rightLeg));
    if (dCut <= dLeft && dCut <= dRight) {
        const V2 dir = -o1u;
        CA_SET_LINE(leftCut + R * invTO * mk(-dir.y, dir.x), dir);
        return true;
    } else if (dLeft <= dRight) {
        if (leftForeign) return false;
        CA_SET_LINE(leftCut + R * invTO * mk(-leftLeg.y, leftLeg.x), leftLeg);
        return true;
    }
    if (rightForeign) return false;
    const V2 dir = -rightLeg;
    CA_SET_LINE(rightCut + R * invTO * mk(-dir.y, dir.x), dir);
    return true;
#undef CA_SET_LINE
}
template <class CoveredFn>
__device__ __forceinline__ bool obst_orca_line(const ObstDev* __restrict__ tab, int e, V2 pos, V2 vel, float R,
                                               float invTO, CoveredFn covered, Line& line) {
    float px = 0.0f, py = 0.0f, dx = 1.0f, dy = 0.0f;
    const bool ok = obst_orca_line4(tab, e, pos, vel, R, invTO, covered, px, py, dx, dy);
    line.point = mk(px, py); line.dir = mk(dx, dy);
    return ok;
}


}  // namespace ca
