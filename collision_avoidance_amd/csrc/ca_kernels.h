// ca_kernels.h -- HIP kernels of the batched collision-avoidance step for gfx950 (MI355X).
//
// Mapping: one LANE per agent, one workgroup per group of arenas (a
// workgroup never splits an arena).  The arena's positions/velocities are staged once into LDS;
// every agent then scans its arena from LDS (broadcast reads), keeps its K nearest neighbours in
// registers, builds its ORCA half-planes into an LDS line table laid out [line][lane] (16 B per
// lane, conflict-free) and solves the 2-D LP over that table.  Arenas are independent, so there
// is no inter-workgroup traffic and no XCD affinity to exploit: the grid is simply arena-major.
//
// Numerics contract: fp32, no FMA contraction (-ffp-contract=off), IEEE sqrt and division, the
// operation order of SURVEY.md Appendix A.  The CPU oracle (oracle/) obeys the same contract, so
// trajectories agree bit for bit.
#pragma once
#include "ca_math.h"
#include <utility>

namespace ca {

constexpr int SMAX = 8;       // CA_MAX_OBST_NEIGHBORS
constexpr float EPS = 0.00001f;

struct ObstDev {  // one obstacle edge (this vertex -> next vertex) with everything ORCA needs about
                  // its two end vertices, so that no dependent `next`/`prev` gathers are required
    float px, py, ux, uy;    // this vertex, unit direction of this edge
    float qx, qy, qux, quy;  // next vertex, unit direction of the edge leaving it
    float pux, puy;          // unit direction of the edge arriving at this vertex (prev's unitDir)
    int next, prev;
    int convex, qconvex, pad0, pad1;
};
static_assert(sizeof(ObstDev) == 64, "edge record is one 64-byte line");

struct Line {
    V2 point, dir;
};

struct StepArgs {
    float *pos_x, *pos_y, *vel_x, *vel_y, *pref_x, *pref_y;
    double *goal_x, *goal_y;        // targets stay fp64 like the reference's Python floats
    const double *goal2_x, *goal2_y;
    float* reward;
    float *orient_x, *orient_y;  // unit vector pos -> goal of the CURRENT state (frame of the observation)
    int *agent_done, *arrive_step, *regoal_count;
    int *nb_count, *nb_idx, *obst_count, *obst_idx;
    int *step_count, *arena_done, *episode;
    unsigned long long* arena_stats;  // [A][8]
    const ObstDev* obst;
    const float* actions;  // null: orca_step
    const float* reset_px; // explicit reset positions (reset kernel only)
    const float* reset_py;
    const int* reset_mask; // [A] reset only the arenas with a non-zero entry (reset kernels only; null = all)
    unsigned long long* dbg;  // CA_STAMPS diagnostic build only: [waves][16] phase cycle counts
    double reward_scale;
    uint64_t seed;
    int64_t arena_offset;
    int n_obst, A, N, P, logP, K, S;
    int a0, a1;  // this launch covers arenas [a0, a1) (chunked launches on several streams)
    uint32_t flags;
    float time_step, neighbor_dist, time_horizon, time_horizon_obst, radius, max_speed;
    int max_step, done_mode;
    float done_x_thresh;
    float spawn_x0, spawn_x1, spawn_y0, spawn_y1, goal_x0, goal_x1, goal_y0, goal_y1;
};

enum { ST_EPISODES = 0, ST_COLL = 1, ST_OBST_COLL = 2, ST_GOALS = 3, ST_OVERFLOW = 4, ST_SUMREW = 5, ST_FROZEN = 6, ST_LASTEP = 7, ST_STRIDE = 8 };

// CA_F_FREEZE: arenas whose arena_done flag is set are left exactly as they are
__device__ __forceinline__ bool arena_frozen(const StepArgs& p, int a) {
    return (p.flags & 16u) != 0 && a < p.a1 && p.arena_done[a] != 0;
}

// ---- line tables ---------------------------------------------------------------------------
struct LdsLines {  // [line][lane] float4 = (point.x, point.y, dir.x, dir.y)
    float4* base;  // already offset by the lane
    int stride;    // lanes per workgroup
    __device__ __forceinline__ Line get(int j) const {
        const float4 v = base[j * stride];
        Line l; l.point = mk(v.x, v.y); l.dir = mk(v.z, v.w);
        return l;
    }
    __device__ __forceinline__ void put(int j, const Line& l) const {
        base[j * stride] = make_float4(l.point.x, l.point.y, l.dir.x, l.dir.y);
    }
};
struct PrivLines {
    const Line* p;
    __device__ __forceinline__ Line get(int j) const { return p[j]; }
};

// App. A.5 LP1.  The contract returns false at the first line that makes the interval empty (or a
// parallel line that excludes it); tLeft only grows and tRight only shrinks, so accumulating the
// same conditions in a flag and finishing the loop gives the same verdict -- and a branch-free body
// whose loads and divisions for consecutive lines overlap (the loop is unrolled by two).
template <class LS>
__device__ __forceinline__ bool lp1(const LS& ls, int lineNo, float radius, V2 opt, bool dirOpt, V2& result) {
    const Line L = ls.get(lineNo);
    const float dp = dot(L.point, L.dir);
    const float disc = sqr(dp) + sqr(radius) - absSq(L.point);
    if (disc < 0.0f) return false;
    const float sq = sqrtf(disc);
    float tLeft = -dp - sq;
    float tRight = -dp + sq;
    bool failed = false;
    auto clip = [&](const Line& M) {
        const float den = det(L.dir, M.dir);
        const float num = det(M.dir, L.point - M.point);
        const bool par = fabsf(den) <= EPS;
        const float t = num / den;
        const bool right = !par && den >= 0.0f, left = !par && !(den >= 0.0f);
        tRight = (right && t < tRight) ? t : tRight;
        tLeft = (left && tLeft < t) ? t : tLeft;
        failed = failed || (par ? (num < 0.0f) : (tLeft > tRight));
    };
    int j = 0;
    for (; j + 1 < lineNo; j += 2) {
        const Line M0 = ls.get(j), M1 = ls.get(j + 1);
        clip(M0);
        clip(M1);
    }
    if (j < lineNo) clip(ls.get(j));
    if (failed) return false;
    if (dirOpt) {
        if (dot(opt, L.dir) > 0.0f) result = L.point + tRight * L.dir;
        else result = L.point + tLeft * L.dir;
    } else {
        const float t = dot(L.dir, opt - L.point);
        if (t < tLeft) result = L.point + tLeft * L.dir;
        else if (t > tRight) result = L.point + tRight * L.dir;
        else result = L.point + t * L.dir;
    }
    return true;
}

// App. A.5 LP2
template <class LS>
__device__ __forceinline__ int lp2(const LS& ls, int n, float radius, V2 opt, bool dirOpt, V2& result) {
    if (dirOpt) result = opt * radius;
    else if (absSq(opt) > sqr(radius)) result = normalize(opt) * radius;
    else result = opt;
    for (int i = 0; i < n; ++i) {
        const Line L = ls.get(i);
        if (det(L.dir, L.point - result) > 0.0f) {
            const V2 tmp = result;
            if (!lp1(ls, i, radius, opt, dirOpt, result)) {
                result = tmp;
                return i;
            }
        }
    }
    return n;
}

// App. A.5 LP3: only for lanes whose LP2 was infeasible -- which in a dense crowd is ~9 % of the
// agent-steps, i.e. a few lanes of EVERY wave.  The projected lines live in private memory: a second
// LDS table for them halves the occupancy of this LDS-bound kernel and was measured slower
// (profiles/r01_k_lp3_lds_negative_result.txt).
template <int MAXL>
__device__ __noinline__ void lp3(LdsLines ls, int n, int numObst, int begin, float radius, V2& result) {
    Line proj[MAXL];
    float distance = 0.0f;
    for (int i = begin; i < n; ++i) {
        const Line Li = ls.get(i);
        if (det(Li.dir, Li.point - result) > distance) {
            int m = 0;
            for (int j = 0; j < numObst; ++j) proj[m++] = ls.get(j);
            for (int j = numObst; j < i; ++j) {
                const Line Lj = ls.get(j);
                Line l;
                const float d = det(Li.dir, Lj.dir);
                if (fabsf(d) <= EPS) {
                    if (dot(Li.dir, Lj.dir) > 0.0f) continue;
                    l.point = 0.5f * (Li.point + Lj.point);
                } else {
                    l.point = Li.point + (det(Lj.dir, Li.point - Lj.point) / d) * Li.dir;
                }
                l.dir = normalize(Lj.dir - Li.dir);
                proj[m++] = l;
            }
            const V2 tmp = result;
            PrivLines pl; pl.p = proj;
            if (lp2(pl, m, radius, mk(-Li.dir.y, Li.dir.x), true, result) < m) result = tmp;
            distance = det(Li.dir, Li.point - result);
        }
    }
}

// ---- ORCA lines in REGISTERS (the fast path of the solve kernel) ---------------------------------
// Slots [0, ST) hold this lane's obstacle lines (the first `no` are valid), slots [ST, ST+KMAX) the
// line of neighbour k (valid for k < ncnt).  Slot order is the contract's line order, so LP2/LP1 run
// over the slots with every loop fully unrolled: all indices are compile-time constants, the table
// lives in VGPRs and the clipping loop has no memory latency at all.
__device__ __forceinline__ float4 pack_line(const Line& l) { return make_float4(l.point.x, l.point.y, l.dir.x, l.dir.y); }
__device__ __forceinline__ Line unpack_line(const float4& v) { Line l; l.point = mk(v.x, v.y); l.dir = mk(v.z, v.w); return l; }

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>); every index is
// a constant expression from the start, so the slot array is promoted to registers
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int ML, int ST, int I>
__device__ __forceinline__ bool lp1_reg(const float4 (&L)[ML], int no, float radius, V2 opt, V2& result) {
    const Line Li = unpack_line(L[I]);
    const float dp = dot(Li.point, Li.dir);
    const float disc = sqr(dp) + sqr(radius) - absSq(Li.point);
    if (disc < 0.0f) return false;
    const float sq = sqrtf(disc);
    float tLeft = -dp - sq;
    float tRight = -dp + sq;
    bool failed = false;
    static_for<I>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if (j >= ST || j < no) {  // an earlier line that exists
            const Line M = unpack_line(L[j]);
            const float den = det(Li.dir, M.dir);
            const float num = det(M.dir, Li.point - M.point);
            const bool par = fabsf(den) <= EPS;
            const float t = num / den;
            const bool right = !par && den >= 0.0f, left = !par && !(den >= 0.0f);
            tRight = (right && t < tRight) ? t : tRight;
            tLeft = (left && tLeft < t) ? t : tLeft;
            failed = failed || (par ? (num < 0.0f) : (tLeft > tRight));
        }
    });
    if (failed) return false;
    const float t = dot(Li.dir, opt - Li.point);
    if (t < tLeft) result = Li.point + tLeft * Li.dir;
    else if (t > tRight) result = Li.point + tRight * Li.dir;
    else result = Li.point + t * Li.dir;
    return true;
}

// App. A.5 LP2 (dirOpt = false) over the register slots; returns the contract's line index of the
// first infeasible line, or the line count when all lines are satisfied.
template <int ML, int ST>
__device__ __forceinline__ int lp2_reg(const float4 (&L)[ML], int no, int ncnt, float radius, V2 opt, V2& result) {
    if (absSq(opt) > sqr(radius)) result = normalize(opt) * radius;
    else result = opt;
    int fail = no + ncnt;
    bool alive = true;
    static_for<ML>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const bool valid = (i < ST) ? (i < no) : (i - ST < ncnt);
        if (alive && valid) {
            const Line Li = unpack_line(L[i]);
            if (det(Li.dir, Li.point - result) > 0.0f) {
                const V2 tmp = result;
                if (!lp1_reg<ML, ST, i>(L, no, radius, opt, result)) {
                    result = tmp;
                    fail = (i < ST) ? i : no + (i - ST);
                    alive = false;
                }
            }
        }
    });
    return fail;
}

// Orders the LDS traffic of ONE wave: LDS executes a wave's instructions in issue order, so lanes of
// the same wave only need the compiler not to move accesses across this point and the earlier
// operations to have been issued and returned (s_waitcnt lgkmcnt(0)).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int POOL_SLOTS = 16;  // LP3 pool slots per wave (lanes beyond that take another round)

// App. A.5 LP3, FOUR LANES PER AGENT.  LP3 is needed by ~9 % of the agents of a dense crowd, i.e. by five or six
// lanes of every wave, while it is the longest dependent computation of the step: solved one agent per lane it
// keeps a wave busy at a tenth of its width.  Here the agents that need it sit in the wave's LDS pool (lines,
// projected lines and a header per slot) and lanes 4 s .. 4 s + 3 work for slot s:
//   * the projected lines of a violated line i are built four at a time and compacted in order (ballot rank);
//   * LP2 over them stays sequential, but each LP1 inside it -- the clipping of line ii against the ii lines
//     before it -- is dealt to the four lanes and merged: tLeft is a maximum, tRight a minimum and the failure
//     flag an OR of per-line conditions, all independent of the order, so the merged values are bit for bit
//     those of the serial loop (whose running `tLeft > tRight` test equals the test on the final values,
//     because tLeft only grows and tRight only shrinks).
// Every lane of a group holds the same `result`; arithmetic per line is that of lp1()/lp3() above.
// header of slot s: pool[(2 ML - 1) * POOL_SLOTS + s] = (result.x, result.y, bits(n | numObst << 8 | begin << 16), -)
__device__ __noinline__ void lp3_coop(float4* pool, int ML, int nslots, float radius) {
    const int lane = threadIdx.x & 63, slot = lane >> 2, q = lane & 3;
    float4* hdr = pool + (size_t)(2 * ML - 1) * POOL_SLOTS;
    LdsLines ls; ls.base = pool + slot; ls.stride = POOL_SLOTS;
    LdsLines pj; pj.base = pool + (size_t)ML * POOL_SLOTS + slot; pj.stride = POOL_SLOTS;
    const bool live = slot < nslots;
    const float4 h = live ? hdr[slot] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int packed = __float_as_int(h.z);
    const int n = live ? (packed & 0xFF) : 0, numObst = (packed >> 8) & 0xFF, begin = (packed >> 16) & 0xFF;
    const int gshift = lane & ~3;
    V2 result = mk(h.x, h.y);
    float distance = 0.0f;
    for (int i = begin; i < n; ++i) {
        const Line Li = ls.get(i);
        if (det(Li.dir, Li.point - result) > distance) {
            for (int j = q; j < numObst; j += 4) pj.put(j, ls.get(j));
            int m = numObst;
            for (int j0 = numObst; j0 < i; j0 += 4) {
                const int j = j0 + q;
                bool valid = j < i;
                Line l; l.point = mk(0.0f, 0.0f); l.dir = mk(1.0f, 0.0f);
                if (valid) {
                    const Line Lj = ls.get(j);
                    const float d = det(Li.dir, Lj.dir);
                    if (fabsf(d) <= EPS) {
                        if (dot(Li.dir, Lj.dir) > 0.0f) valid = false;
                        else l.point = 0.5f * (Li.point + Lj.point);
                    } else {
                        l.point = Li.point + (det(Lj.dir, Li.point - Lj.point) / d) * Li.dir;
                    }
                    l.dir = normalize(Lj.dir - Li.dir);
                }
                const unsigned mask = (unsigned)(__ballot(valid) >> gshift) & 0xFu;
                if (valid) pj.put(m + __popc(mask & ((1u << q) - 1u)), l);
                m += __popc(mask);
            }
            wave_lds_sync();
            const V2 opt = mk(-Li.dir.y, Li.dir.x);
            V2 res = opt * radius;  // lp2(..., dirOpt = true)
            bool ok = true;
            for (int ii = 0; ii < m && ok; ++ii) {
                const Line L = pj.get(ii);
                if (det(L.dir, L.point - res) > 0.0f) {
                    const float dp = dot(L.point, L.dir);
                    const float disc = sqr(dp) + sqr(radius) - absSq(L.point);
                    int failed = disc < 0.0f ? 1 : 0;
                    const float sq = sqrtf(disc);
                    float tLeft = -dp - sq;
                    float tRight = -dp + sq;
                    for (int jj = q; jj < ii; jj += 4) {
                        const Line M = pj.get(jj);
                        const float den = det(L.dir, M.dir);
                        const float num = det(M.dir, L.point - M.point);
                        const bool par = fabsf(den) <= EPS;
                        const float t = num / den;
                        const bool right = !par && den >= 0.0f, left = !par && !(den >= 0.0f);
                        tRight = (right && t < tRight) ? t : tRight;
                        tLeft = (left && tLeft < t) ? t : tLeft;
                        failed |= (par && num < 0.0f) ? 1 : 0;
                    }
#pragma unroll
                    for (int x = 1; x <= 2; x <<= 1) {
                        const float oR = __shfl_xor(tRight, x), oL = __shfl_xor(tLeft, x);
                        tRight = (oR < tRight) ? oR : tRight;
                        tLeft = (tLeft < oL) ? oL : tLeft;
                        failed |= __shfl_xor(failed, x);
                    }
                    if (failed || tLeft > tRight) ok = false;  // lp2 stops here and LP3 keeps its previous result
                    else res = (dot(opt, L.dir) > 0.0f) ? L.point + tRight * L.dir : L.point + tLeft * L.dir;
                }
            }
            if (ok) result = res;
            distance = det(Li.dir, Li.point - result);
            wave_lds_sync();  // the projected lines are rewritten for the next violated line
        }
    }
    if (live && q == 0) hdr[slot] = make_float4(result.x, result.y, h.z, 0.0f);
}

// App. A.4: the half-plane induced by one neighbouring agent (both agents have radius R)
__device__ __forceinline__ Line agent_orca_line(V2 pos, V2 vel, V2 opos, V2 ovel, float R, float invT, float invDt) {
    const V2 rp = opos - pos;
    const V2 rv = vel - ovel;
    const float distSq = absSq(rp);
    const float cr = R + R;
    const float crSq = sqr(cr);
    Line line;
    V2 u;
    if (distSq > crSq) {
        const V2 w = rv - invT * rp;
        const float wLenSq = absSq(w);
        const float dp1 = dot(w, rp);
        if (dp1 < 0.0f && sqr(dp1) > crSq * wLenSq) {
            const float wLen = sqrtf(wLenSq);
            const V2 unitW = vdiv(w, wLen);
            line.dir = mk(unitW.y, -unitW.x);
            u = (cr * invT - wLen) * unitW;
        } else {
            const float leg = sqrtf(distSq - crSq);
            if (det(rp, w) > 0.0f)
                line.dir = vdiv(mk(rp.x * leg - rp.y * cr, rp.x * cr + rp.y * leg), distSq);
            else
                line.dir = -vdiv(mk(rp.x * leg + rp.y * cr, -rp.x * cr + rp.y * leg), distSq);
            const float dp2 = dot(rv, line.dir);
            u = dp2 * line.dir - rv;
        }
    } else {
        const V2 w = rv - invDt * rp;
        const float wLen = vabs(w);
        const V2 unitW = vdiv(w, wLen);
        line.dir = mk(unitW.y, -unitW.x);
        u = (cr * invDt - wLen) * unitW;
    }
    line.point = vel + 0.5f * u;
    return line;
}

__device__ __forceinline__ ObstDev load_obst(const ObstDev* __restrict__ t, int i) {
    const int4* q = reinterpret_cast<const int4*>(t + i);
    const int4 a = q[0], b = q[1], c = q[2], d = q[3];
    ObstDev o;
    o.px = __int_as_float(a.x); o.py = __int_as_float(a.y); o.ux = __int_as_float(a.z); o.uy = __int_as_float(a.w);
    o.qx = __int_as_float(b.x); o.qy = __int_as_float(b.y); o.qux = __int_as_float(b.z); o.quy = __int_as_float(b.w);
    o.pux = __int_as_float(c.x); o.puy = __int_as_float(c.y); o.next = c.z; o.prev = c.w;
    o.convex = d.x; o.qconvex = d.y; o.pad0 = 0; o.pad1 = 0;
    return o;
}

// App. A.3: the half-plane induced by the obstacle edge e.  Returns false when the edge yields no
// line (already covered, non-convex vertex, foreign leg).  "o1"/"o2" are the edge's two vertices;
// the oblique cases collapse the edge onto one of them, exactly as the contract's o2<-o1 / o1<-o2.
// `covered(a, b)`: true when some earlier obstacle line already excludes both scaled end points.
template <class CoveredFn>
__device__ __forceinline__ bool obst_orca_line(const ObstDev* __restrict__ tab, int e, V2 pos, V2 vel, float R,
                                               float invTO, CoveredFn covered, Line& line) {
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
    const float s = dot(-rp1, ov) / absSq(ov);
    const float distSqLine = absSq(-rp1 - s * ov);
    if (s < 0.0f && distSq1 <= radiusSq) {
        if (o1c) {
            line.point = mk(0.0f, 0.0f);
            line.dir = normalize(mk(-rp1.y, rp1.x));
            return true;
        }
        return false;
    } else if (s > 1.0f && distSq2 <= radiusSq) {
        if (o2c && det(rp2, o2u) >= 0.0f) {
            line.point = mk(0.0f, 0.0f);
            line.dir = normalize(mk(-rp2.y, rp2.x));
            return true;
        }
        return false;
    } else if (s >= 0.0f && s < 1.0f && distSqLine <= radiusSq) {
        line.point = mk(0.0f, 0.0f);
        line.dir = -o1u;
        return true;
    }
    V2 leftLeg, rightLeg;
    if (s < 0.0f && distSqLine <= radiusSq) {
        if (!o1c) return false;
        o2p = o1p; o2u = o1u; o2c = o1c; same = true;  // o2 <- o1
        const float leg1 = sqrtf(distSq1 - radiusSq);
        leftLeg = vdiv(mk(rp1.x * leg1 - rp1.y * R, rp1.x * R + rp1.y * leg1), distSq1);
        rightLeg = vdiv(mk(rp1.x * leg1 + rp1.y * R, -rp1.x * R + rp1.y * leg1), distSq1);
    } else if (s > 1.0f && distSqLine <= radiusSq) {
        if (!o2c) return false;
        lnu = o1u;                                     // the new o1's prev vertex is the old o1
        o1p = o2p; o1u = o2u; o1c = o2c; same = true;  // o1 <- o2
        const float leg2 = sqrtf(distSq2 - radiusSq);
        leftLeg = vdiv(mk(rp2.x * leg2 - rp2.y * R, rp2.x * R + rp2.y * leg2), distSq2);
        rightLeg = vdiv(mk(rp2.x * leg2 + rp2.y * R, -rp2.x * R + rp2.y * leg2), distSq2);
    } else {
        if (o1c) {
            const float leg1 = sqrtf(distSq1 - radiusSq);
            leftLeg = vdiv(mk(rp1.x * leg1 - rp1.y * R, rp1.x * R + rp1.y * leg1), distSq1);
        } else {
            leftLeg = -o1u;
        }
        if (o2c) {
            const float leg2 = sqrtf(distSq2 - radiusSq);
            rightLeg = vdiv(mk(rp2.x * leg2 + rp2.y * R, -rp2.x * R + rp2.y * leg2), distSq2);
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
    const float t = same ? 0.5f : dot(vel - leftCut, cutVec) / absSq(cutVec);
    const float tLeft = dot(vel - leftCut, leftLeg);
    const float tRight = dot(vel - rightCut, rightLeg);
    if ((t < 0.0f && tLeft < 0.0f) || (same && tLeft < 0.0f && tRight < 0.0f)) {
        const V2 unitW = normalize(vel - leftCut);
        line.dir = mk(unitW.y, -unitW.x);
        line.point = leftCut + R * invTO * unitW;
        return true;
    } else if (t > 1.0f && tRight < 0.0f) {
        const V2 unitW = normalize(vel - rightCut);
        line.dir = mk(unitW.y, -unitW.x);
        line.point = rightCut + R * invTO * unitW;
        return true;
    }
    const float INF = __int_as_float(0x7f800000);
    const float dCut = (t < 0.0f || t > 1.0f || same) ? INF : absSq(vel - (leftCut + t * cutVec));
    const float dLeft = (tLeft < 0.0f) ? INF : absSq(vel - (leftCut + tLeft * leftLeg));
    const float dRight = (tRight < 0.0f) ? INF : absSq(vel - (rightCut + tRight * rightLeg));
    if (dCut <= dLeft && dCut <= dRight) {
        line.dir = -o1u;
        line.point = leftCut + R * invTO * mk(-line.dir.y, line.dir.x);
        return true;
    } else if (dLeft <= dRight) {
        if (leftForeign) return false;
        line.dir = leftLeg;
        line.point = leftCut + R * invTO * mk(-line.dir.y, line.dir.x);
        return true;
    }
    if (rightForeign) return false;
    line.dir = -rightLeg;
    line.point = rightCut + R * invTO * mk(-line.dir.y, line.dir.x);
    return true;
}

// Sorted insertion into a register-resident list kept ascending, the last entry falling off.
// An entry is the 64-bit key (distance bits << 32 | index) held in a double register pair: for
// non-negative floats the bit pattern is monotone, so key order IS the (distance, index) order of the
// contract (App. A.2: ascending distance, ties to the lower index), and as positive, never-NaN doubles
// the keys are ordered by v_min_f64 / v_max_f64.  Insertion is then, for every slot independently and
// in place,   new[k] = max(old[k-1], min(old[k], x))   -- two VALU instructions per slot, no compare
// masks, no register copies.  (Inline asm because the compiler would add a canonicalising
// v_max_f64 v,v,v per operand; keys are never NaN so nothing needs quieting.)
// A list shorter than the array is stored RIGHT-ALIGNED behind dummy -inf slots (which never move):
// its largest key is then always the last element, a compile-time index.
__device__ __forceinline__ double key_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double key_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double make_key(float d, int idx) {
    return __longlong_as_double((long long)(((unsigned long long)__float_as_uint(d) << 32) | (unsigned)idx));
}
__device__ __forceinline__ float key_dist(double k) { return __uint_as_float((unsigned)((unsigned long long)__double_as_longlong(k) >> 32)); }
__device__ __forceinline__ int key_index(double k) { return (int)(unsigned)(unsigned long long)__double_as_longlong(k); }
template <int MAXN>
__device__ __forceinline__ void sorted_insert(double (&key)[MAXN], double x) {
#pragma unroll
    for (int k = MAXN - 1; k >= 1; --k) key[k] = key_max(key[k - 1], key_min(key[k], x));
    key[0] = key_min(key[0], x);
}
#ifdef CA_STAMPS  // diagnostic build: per-wave cycle count of each phase (never in the product library)
#if CA_STAMPS == 2   // wall-clock variant: the 100 MHz device-wide counter (wave timelines across CUs)
#define CA_STAMP_CLOCK() __builtin_amdgcn_s_memrealtime()
#else                // per-CU shader-clock counter (phase shares inside a wave)
#define CA_STAMP_CLOCK() __builtin_amdgcn_s_memtime()
#endif
#define CA_STAMP(k)                                                                      \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        const unsigned long long _t = CA_STAMP_CLOCK();                                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                              \
        if ((threadIdx.x & 63) == 0 && p.dbg)                                            \
            p.dbg[((size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6)) * 16 + (k)] = _t; \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define CA_STAMP(k) do { } while (0)
#endif
// ============================================================================================
// Neighbour search for every agent (App. A.2): the obstacle edges within range and
// the K nearest agents, written as the lists [A,S,N] / [A,K,N] that the solve kernel and the
// observation read.  A kernel of its own because it needs almost no LDS (the arena's positions,
// 8 B per lane): it runs at full occupancy and is issue-bound, whereas the solve kernel is tied to
// its 16 B x (K+S) line table per lane.
// ============================================================================================
template <int KMAX, int BS>
__device__ __forceinline__ void nbr_body(const StepArgs& p) {
#ifndef CA_NBR_NO_VGPR_PAD
    // Claim 128 VGPRs (the kernel needs 56): at most 4 waves then fit on a SIMD, so a launch that brings one
    // wave per SIMD slot (4096 x 64 lanes on 256 CUs) is spread evenly.  Without it the dispatcher puts
    // anything from 1 to 7 of these light waves on a SIMD and the kernel waits for the fullest one.
    asm volatile("" ::: "v127");
#endif
    __shared__ float s_px[BS];
    __shared__ float s_py[BS];
    const int tid = threadIdx.x;
    const int P = p.P;
    const int la = tid >> p.logP;
    const int i = tid & (P - 1);
    const int apb = BS >> p.logP;
    const int a = p.a0 + blockIdx.x * apb + la;
    const bool active = (a < p.a1) && (i < p.N) && !arena_frozen(p, a);
    const int N = p.N, K = p.K, S = p.S;
    const int q = active ? a * N + i : 0;
    const int lbase = la << p.logP;
    CA_STAMP(12);
    V2 pos = mk(0.0f, 0.0f);
    if (active) pos = mk(p.pos_x[q], p.pos_y[q]);
    s_px[tid] = pos.x; s_py[tid] = pos.y;
    __syncthreads();

    const float INF = __int_as_float(0x7f800000);
    // ---- obstacle neighbours (App. A.2): brute force over the edge table ----
    const int sofs = SMAX - S;  // the S-entry list is right-aligned in the register array
    const double KEY_EMPTY = __longlong_as_double(0x7F800000FFFFFFFFll);  // (+inf, -1)
    const double KEY_DUMMY = __longlong_as_double((long long)0xFFF0000000000000ull);  // -inf: never moves
    double okey[SMAX];
#pragma unroll
    for (int k = 0; k < SMAX; ++k) okey[k] = (k < sofs) ? KEY_DUMMY : KEY_EMPTY;
    int oin = 0;
    {
        const float rangeSq = sqr(p.time_horizon_obst * p.max_speed + p.radius);
        for (int e = 0; e < p.n_obst; ++e) {
            const ObstDev o1 = p.obst[e];
            const V2 a1 = mk(o1.px, o1.py), a2 = mk(o1.qx, o1.qy);
            const float alol = leftOf(a1, a2, pos);
            const float dsl = sqr(alol) / absSq(a2 - a1);
            if (active && dsl < rangeSq && alol < 0.0f) {
                const float dsq = distSqPointSegment(a1, a2, pos);
                if (dsq < rangeSq) {
                    ++oin;
                    sorted_insert<SMAX>(okey, make_key(dsq, e));
                }
            }
        }
    }
    const int ocnt = oin < S ? oin : S;
    CA_STAMP(13);

    // ---- agent neighbours (App. A.2): K nearest within neighbor_dist, ties -> lower index ----
    const int kofs = KMAX - K;  // the K-entry list is right-aligned in the register array
    double nkey[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) nkey[k] = (k < kofs) ? KEY_DUMMY : KEY_EMPTY;
    int ncnt = 0;
    bool scanned = false;
    if constexpr (BS >= 256) {
        // Large arenas (one arena per workgroup, >= 192 agents): a uniform grid with cells at least neighbor_dist
        // wide, rebuilt in LDS every step (counting sort of the agent indices by cell), so that an agent scans
        // the 3 x 3 cells around it -- three contiguous runs of the sorted list -- instead of the whole arena.
        // The cells are visited in no particular index order, so a candidate enters on `distance <= current
        // K-th distance` and the 64-bit (distance, index) keys settle ties; the list is the same K smallest
        // keys within neighbor_dist that the index-order scan keeps.
        if (P == BS && N >= 192 && K > 0) {
            __shared__ unsigned s_box[4];          // ordered-uint images of min x, min y, max x, max y
            __shared__ int s_ccnt[256];            // agents per cell
            __shared__ int s_cstart[257];          // first position of a cell in s_sorted
            __shared__ unsigned short s_sorted[BS];
            auto ord = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
            auto unord = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); };
            if (tid < 2) s_box[tid] = 0xFFFFFFFFu;
            if (tid >= 2 && tid < 4) s_box[tid] = 0u;
            if (tid < 256) s_ccnt[tid] = 0;
            __syncthreads();
            const bool in_arena = (a < p.a1) && (i < N);  // frozen arenas skip the scan but keep the barriers
            if (in_arena) {
                atomicMin(&s_box[0], ord(pos.x)); atomicMin(&s_box[1], ord(pos.y));
                atomicMax(&s_box[2], ord(pos.x)); atomicMax(&s_box[3], ord(pos.y));
            }
            __syncthreads();
            const float x0 = unord(s_box[0]), y0 = unord(s_box[1]);
            const float ex = unord(s_box[2]) - x0, ey = unord(s_box[3]) - y0;
            const float cs = fmaxf(p.neighbor_dist, fmaxf(ex, ey) * (1.0f / 15.5f));  // at most 16 x 16 cells
            const float ics = 1.0f / cs;
            const int Gx = min(16, (int)(ex * ics) + 1), Gy = min(16, (int)(ey * ics) + 1);
            const int cx = min(Gx - 1, max(0, (int)((pos.x - x0) * ics))), cy = min(Gy - 1, max(0, (int)((pos.y - y0) * ics)));
            int rank = 0;
            if (in_arena) rank = atomicAdd(&s_ccnt[cy * Gx + cx], 1);
            __syncthreads();
            if (tid < 64) {  // exclusive prefix sum over the (<= 256) cells: four cells per lane of the first wave
                const int c0 = s_ccnt[4 * tid], c1 = s_ccnt[4 * tid + 1], c2 = s_ccnt[4 * tid + 2], c3 = s_ccnt[4 * tid + 3];
                int incl = c0 + c1 + c2 + c3;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off);
                    if (tid >= off) incl += t;
                }
                const int b = incl - (c0 + c1 + c2 + c3);
                s_cstart[4 * tid] = b; s_cstart[4 * tid + 1] = b + c0; s_cstart[4 * tid + 2] = b + c0 + c1;
                s_cstart[4 * tid + 3] = b + c0 + c1 + c2;
                if (tid == 63) s_cstart[256] = incl;
            }
            __syncthreads();
            if (in_arena) s_sorted[s_cstart[cy * Gx + cx] + rank] = (unsigned short)i;
            __syncthreads();
            const float rangeSq0 = sqr(p.neighbor_dist);
            float rangeK = rangeSq0;  // distance of the current K-th entry once the list is full
            for (int ry = -1; ry <= 1; ++ry) {
                const int row = cy + ry;
                int lo = 0, hi = 0;
                if (active && row >= 0 && row < Gy) {
                    lo = s_cstart[row * Gx + max(cx - 1, 0)];
                    hi = s_cstart[row * Gx + min(cx + 1, Gx - 1) + 1];
                }
                for (int t = lo; t < hi; ++t) {
                    const int j = s_sorted[t];
                    const float dsq = absSq(pos - mk(s_px[j], s_py[j]));
                    if (j != i && dsq < rangeSq0 && dsq <= rangeK) {
                        sorted_insert<KMAX>(nkey, make_key(dsq, j));
                        if (ncnt < K) ++ncnt;
                        if (ncnt == K) rangeK = key_dist(nkey[KMAX - 1]);
                    }
                }
            }
            scanned = true;
        }
    }
    if (K > 0 && !scanned) {
        float rangeSq = sqr(p.neighbor_dist);
        V2 o_next = mk(s_px[lbase], s_py[lbase]);
        for (int j = 0; j < N; ++j) {
            const V2 o = o_next;  // the next candidate's position is in flight while this one is inserted
            if (j + 1 < N) o_next = mk(s_px[lbase + j + 1], s_py[lbase + j + 1]);
            const float dsq = absSq(pos - o);
            if (active && j != i && dsq < rangeSq) {
                sorted_insert<KMAX>(nkey, make_key(dsq, j));
                if (ncnt < K) ++ncnt;
                if (ncnt == K) rangeSq = key_dist(nkey[KMAX - 1]);
            }
        }
    }

    CA_STAMP(14);
    if (active) {
        if (oin > S) atomicAdd(reinterpret_cast<int*>(&p.arena_stats[(size_t)a * ST_STRIDE + ST_OVERFLOW]), 1);
        p.nb_count[q] = ncnt;
        p.obst_count[q] = ocnt;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k >= kofs) p.nb_idx[((size_t)a * K + (k - kofs)) * N + i] = key_index(nkey[k]);
#pragma unroll
        for (int k = 0; k < SMAX; ++k)
            if (k >= sofs) p.obst_idx[((size_t)a * S + (k - sofs)) * N + i] = key_index(okey[k]);
    }
    CA_STAMP(15);
}

template <int KMAX, int BS>
__global__ __launch_bounds__(BS) void nbr_kernel(const StepArgs p) {
    nbr_body<KMAX, BS>(p);
}

// LDS carve-up of the step kernel (bytes): lines | px py vx vy | misc ints
// ST = 0: the LDS line table; ST > 0 (register lines): per wave an LP3 pool of POOL_SLOTS slots x
// (ML lines + ML - 1 projected lines + a header), ML = ST + KMAX
__host__ __device__ inline size_t step_lds_bytes(int BS, int K, int S, int ST = 0, int KMAX = 0) {
    if (ST > 0) return (size_t)(BS / 64) * (2 * (ST + KMAX)) * POOL_SLOTS * 16 + (size_t)BS * 32;
    return (size_t)BS * ((size_t)(K + S) * 16 + 16 + 16);
}

// ============================================================================================
// One environment step for every arena (SURVEY.md A5/A6 -> A10-A15 -> A16-A18 + A20).
// actions != null : env.py:367-416 `step`;  actions == null : `orca_step` (env.py:447-450,
// ALAN:631-636) followed by the done test of ALAN:118-121 unless CA_F_NODONE.
// ============================================================================================

// ST = 0: ORCA lines in the LDS table [K+S][BS] (any K <= 16, S <= 8).
// ST > 0: ORCA lines in registers (ST obstacle slots + KMAX neighbour slots), LP2/LP1 fully unrolled,
//         LP3 through a small per-wave LDS pool.  Needs S <= ST; ~8 KB of LDS per wave instead of
//         16 KB; built for 4 waves per SIMD (<= 128 VGPRs), i.e. 16 waves per CU: the 4096 arenas
//         of the C3 workload are all resident at once instead of taking 1.6 rounds at 10 per CU.
template <int KMAX, int BS, int ST, bool FUSE>
__global__ __launch_bounds__(BS, ST > 0 ? 4 : 1) void step_kernel(const StepArgs p) {
    extern __shared__ float4 smem4[];
    // FUSE: the neighbour search runs at the head of this kernel instead of in a launch of its own (one
    // drain/fill less per step, and its dispatch skew overlaps useful work).  A lane later reads back only the
    // lists of its own agent, which it wrote itself; the search's LDS arrays are not used again.
    if constexpr (FUSE) nbr_body<KMAX, BS>(p);
    constexpr int ML = ST + KMAX;  // register slots (ST > 0)
    const int tid = threadIdx.x;
    const int P = p.P;
    const int la = tid >> p.logP;
    const int i = tid & (P - 1);
    const int apb = BS >> p.logP;
    const int a = p.a0 + blockIdx.x * apb + la;
    const bool frozen = arena_frozen(p, a);  // CA_F_FREEZE: the episode of this arena is over
    const bool active = (a < p.a1) && (i < p.N) && !frozen;
    if (frozen && i == 0) p.arena_stats[(size_t)a * ST_STRIDE + ST_FROZEN] += 1;
    const int N = p.N, K = p.K, S = p.S;
    const int q = active ? a * N + i : 0;
    const int lbase = la << p.logP;

    float4* s_lines = smem4;  // ST = 0: [(K+S)][BS];  ST > 0: [waves][2 ML][POOL_SLOTS] (last row: slot headers)
    float* s_px = reinterpret_cast<float*>(
        smem4 + (ST > 0 ? (size_t)(BS / 64) * (2 * ML) * POOL_SLOTS : (size_t)(K + S) * BS));
    float* s_py = s_px + BS;
    float* s_vx = s_py + BS;
    float* s_vy = s_vx + BS;
    int* s_misc = reinterpret_cast<int*>(s_vy + BS);            // [BS][4]
    LdsLines ls; ls.base = s_lines + tid; ls.stride = BS;       // (ST = 0 only)

    CA_STAMP(0);
    // ---- load own state (coalesced SoA) ----
    V2 pos = mk(0.0f, 0.0f), vel = mk(0.0f, 0.0f), pref = mk(0.0f, 0.0f);
    double gx = 0.0, gy = 0.0;
    int done = 1;
    double pf_x = 1.0, pf_y = 0.0, rl_x = 1.0, rl_y = 0.0;
    if (active) {
        pos = mk(p.pos_x[q], p.pos_y[q]);
        vel = mk(p.vel_x[q], p.vel_y[q]);
        gx = p.goal_x[q]; gy = p.goal_y[q];
        done = p.agent_done[q];
        if (p.actions) {  // env.py:371-383
            pref_dir64(pos.x, pos.y, gx, gy, &pf_x, &pf_y);
            double sn, cs;
            sincos64((double)p.actions[q], &sn, &cs);
            rl_x = pf_x * cs - pf_y * sn;
            rl_y = pf_x * sn + pf_y * cs;
            pref = mk((float)rl_x, (float)rl_y);
        } else {
            pref = mk(p.pref_x[q], p.pref_y[q]);
        }
    }
    s_px[tid] = pos.x; s_py[tid] = pos.y; s_vx[tid] = vel.x; s_vy[tid] = vel.y;
    __syncthreads();

    CA_STAMP(1);
    // ---- neighbour lists of this step (App. A.2), produced by nbr_kernel ----
    const int ocnt = active ? p.obst_count[q] : 0;
    const int ncnt = active ? p.nb_count[q] : 0;
    CA_STAMP(2);
    CA_STAMP(3);
    const float R = p.radius;
    V2 nv = mk(0.0f, 0.0f);
    if constexpr (ST > 0) {
        // ================= register path =================
        float4 L[ML];
        static_for<ML>([&](auto kc) __attribute__((always_inline)) { L[decltype(kc)::value] = make_float4(0.0f, 0.0f, 1.0f, 0.0f); });
        int no = 0;  // obstacle lines produced so far (slots [0, no))
        {
            const float invTO = 1.0f / p.time_horizon_obst;
            int e_next = (ocnt > 0) ? p.obst_idx[((size_t)a * S + 0) * N + i] : 0;
            for (int s = 0; s < S; ++s) {
                if (s < ocnt) {
                    const int e = e_next;
                    if (s + 1 < ocnt) e_next = p.obst_idx[((size_t)a * S + (s + 1)) * N + i];
                    auto covered = [&](V2 c1, V2 c2) __attribute__((always_inline)) {
                        bool c = false;
                        static_for<ST>([&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            const Line M = unpack_line(L[j]);
                            if (j < no && det(c1 - M.point, M.dir) - invTO * R >= -EPS &&
                                det(c2 - M.point, M.dir) - invTO * R >= -EPS)
                                c = true;
                        });
                        return c;
                    };
                    Line line;
                    if (obst_orca_line(p.obst, e, pos, vel, R, invTO, covered, line)) {
                        const float4 pl = pack_line(line);
                        static_for<ST>([&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            if (j == no) L[j] = pl;  // (a ?: on the struct type would select between addresses)
                        });
                        ++no;
                    }
                }
            }
        }
        CA_STAMP(4);
        {
            const float invT = 1.0f / p.time_horizon;
            const float invDt = 1.0f / p.time_step;
            int jn[KMAX];  // all neighbour indices in flight at once
            static_for<KMAX>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                jn[k] = (k < ncnt) ? p.nb_idx[((size_t)a * K + k) * N + i] : 0;
            });
            static_for<KMAX>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                if (k < ncnt) {
                    const int j = lbase + jn[k];
                    L[ST + k] = pack_line(agent_orca_line(pos, vel, mk(s_px[j], s_py[j]), mk(s_vx[j], s_vy[j]), R, invT, invDt));
                }
            });
        }
        CA_STAMP(5);
        // ---- 2-D linear program (App. A.5) on the register slots ----
        const int nl = no + ncnt;
        int fail = nl;
        if (active) fail = lp2_reg<ML, ST>(L, no, ncnt, p.max_speed, pref, nv);
        CA_STAMP(6);
        // ---- LP3 for the lanes whose LP2 was infeasible: they copy their lines into a slot of the
        // wave's LDS pool and solve there; more than POOL_SLOTS such lanes take further rounds ----
        {
            float4* pool = s_lines + (size_t)(tid >> 6) * (2 * ML) * POOL_SLOTS;
            float4* hdr = pool + (size_t)(2 * ML - 1) * POOL_SLOTS;
            bool need = active && fail < nl;
            const unsigned long long below = (1ull << (tid & 63)) - 1ull;
            while (true) {
                const unsigned long long m = __ballot(need);
                if (!m) break;
                const int rank = __popcll(m & below);
                const bool mine = need && rank < POOL_SLOTS;
                if (mine) {
                    LdsLines pls; pls.base = pool + rank; pls.stride = POOL_SLOTS;
                    static_for<ML>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k = decltype(kc)::value;
                        const bool valid = (k < ST) ? (k < no) : (k - ST < ncnt);
                        if (valid) pls.base[((k < ST) ? k : no + (k - ST)) * POOL_SLOTS] = L[k];
                    });
                    hdr[rank] = make_float4(nv.x, nv.y, __int_as_float(nl | (no << 8) | (fail << 16)), 0.0f);
                }
                wave_lds_sync();
                const int waiting = __popcll(m);
                lp3_coop(pool, ML, waiting < POOL_SLOTS ? waiting : POOL_SLOTS, p.max_speed);  // the whole wave works
                wave_lds_sync();
                if (mine) {
                    const float4 h = hdr[rank];
                    nv = mk(h.x, h.y);
                    need = false;
                }
            }
        }
    } else {
        // ================= LDS-table path =================
        int nl = 0;
    {
        const float invTO = 1.0f / p.time_horizon_obst;
        int e_next = (ocnt > 0) ? p.obst_idx[((size_t)a * S + 0) * N + i] : 0;
        for (int s = 0; s < S; ++s) {
            if (s < ocnt) {
                const int e = e_next;
                if (s + 1 < ocnt) e_next = p.obst_idx[((size_t)a * S + (s + 1)) * N + i];
                Line line;
                auto covered = [&](V2 c1, V2 c2) {
                    for (int j = 0; j < nl; ++j) {
                        const Line M = ls.get(j);
                        if (det(c1 - M.point, M.dir) - invTO * R >= -EPS && det(c2 - M.point, M.dir) - invTO * R >= -EPS)
                            return true;
                    }
                    return false;
                };
                if (obst_orca_line(p.obst, e, pos, vel, R, invTO, covered, line)) {
                    ls.put(nl, line);
                    ++nl;
                }
            }
        }
    }
    const int numObstLines = nl;
    CA_STAMP(4);
    {
        const float invT = 1.0f / p.time_horizon;
        const float invDt = 1.0f / p.time_step;
        int j_next = (ncnt > 0) ? p.nb_idx[((size_t)a * K + 0) * N + i] : 0;
        for (int k = 0; k < K; ++k) {
            if (k < ncnt) {
                const int j = lbase + j_next;
                if (k + 1 < ncnt) j_next = p.nb_idx[((size_t)a * K + (k + 1)) * N + i];
                const Line line = agent_orca_line(pos, vel, mk(s_px[j], s_py[j]), mk(s_vx[j], s_vy[j]), R, invT, invDt);
                ls.put(nl, line);
                ++nl;
            }
        }
    }

    CA_STAMP(5);
    // ---- 2-D linear program (App. A.5) ----
    int fail = nl;
    if (active) fail = lp2(ls, nl, p.max_speed, pref, false, nv);
    CA_STAMP(6);
    if (active && fail < nl) lp3<KMAX + SMAX>(ls, nl, numObstLines, fail, p.max_speed, nv);
    }
    if (active) {  // ---- integrate (App. A.1) ----
        vel = nv;
        pos = pos + vel * p.time_step;
    }
    CA_STAMP(7);

    __syncthreads();  // every lane is done with the pre-step arena image
    s_px[tid] = pos.x; s_py[tid] = pos.y;
    s_misc[tid * 4 + 0] = 0; s_misc[tid * 4 + 1] = 0; s_misc[tid * 4 + 2] = 0; s_misc[tid * 4 + 3] = 0;
    __syncthreads();
    int* red = s_misc + la * 4;  // per-arena: [0] not-done agents, [1] pairs, [2] wall hits, [3] goals

    if (active) {
        if (p.flags & 2u) {  // CA_F_STATS
            int pairs = 0;
            const float crSq = sqr(R + R);
            for (int j = i + 1; j < N; ++j)
                if (absSq(pos - mk(s_px[lbase + j], s_py[lbase + j])) < crSq) ++pairs;
            bool wall = false;
            for (int e = 0; e < p.n_obst; ++e) {
                const ObstDev o1 = p.obst[e];
                if (distSqPointSegment(mk(o1.px, o1.py), mk(o1.qx, o1.qy), pos) < sqr(R)) wall = true;
            }
            if (pairs) atomicAdd(&red[1], pairs);
            if (wall) atomicAdd(&red[2], 1);
        }
    }

    CA_STAMP(8);
    // ---- reward (env.py:389-400) or preferred velocity towards the goal (env.py:449) ----
    float rew = 0.0f;
    if (active) {
        if (p.actions) {
            const float scale = (float)p.reward_scale;
            const float r_goal = vel.x * (float)pf_x + vel.y * (float)pf_y;
            const float r_polite = vel.x * (float)rl_x + vel.y * (float)rl_y;
            rew = scale * r_goal + (1.0f - scale) * r_polite;
            p.reward[q] = rew;
        } else {
            double dx, dy;
            pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
            pref = mk((float)dx, (float)dy);
        }
    }

    CA_STAMP(9);
    // ---- step counter and done test (env.py:352-365, 404-410; ALAN:118-121, 547-566) ----
    const bool nodone = (p.flags & 8u) != 0;  // CA_F_NODONE
    bool goal_changed = false;
    int steps = active ? p.step_count[a] : 0;
    if (!p.actions && !nodone) ++steps;
    if (active && !nodone) {
        bool hit = false;
        if (p.done_mode == 0) {
            hit = (done == 0) && (pos.x < p.done_x_thresh);
        } else {
            const double dx = (double)pos.x - gx, dy = (double)pos.y - gy;
            const double lim = 2.0 * (double)p.radius;
            hit = (dx * dx + dy * dy) < lim * lim;
            if (p.done_mode == 1) hit = hit && (done == 0);
        }
        if (hit) {
            if (p.done_mode == 2) {
                const int rc = p.regoal_count[q];
                double u0, u1;
                rng2(p.seed, p.arena_offset + a, i, RNG_REGOAL, (uint32_t)rc, &u0, &u1);
                gx = uniform64((double)p.goal_x0, (double)p.goal_x1, u0);
                gy = uniform64((double)p.goal_y0, (double)p.goal_y1, u1);
                p.regoal_count[q] = rc + 1;
            } else {
                done = 1;
                p.arrive_step[q] = steps;
                gx = p.goal2_x[q]; gy = p.goal2_y[q];
                p.agent_done[q] = 1;
            }
            p.goal_x[q] = gx; p.goal_y[q] = gy;
            goal_changed = true;
            atomicAdd(&red[3], 1);
        }
    }
    if (p.actions) ++steps;
    if (active && done == 0) atomicAdd(&red[0], 1);
    __syncthreads();

    bool all_done = false;
    if (active) {
        all_done = !nodone && (red[0] == 0);
        if (p.max_step > 0 && steps >= p.max_step) all_done = true;
    }
    const bool do_reset = all_done && (p.flags & 4u);  // CA_F_AUTORESET
    int epi = 0;
    if (do_reset) {  // env.py:461-488 for this arena
        epi = p.episode[a];
        double u0, u1;
        rng2(p.seed, p.arena_offset + a, i, RNG_RESET, (uint32_t)epi, &u0, &u1);
        pos = mk((float)uniform64((double)p.spawn_x0, (double)p.spawn_x1, u0),
                 (float)uniform64((double)p.spawn_y0, (double)p.spawn_y1, u1));
        done = 0;
        p.agent_done[q] = 0;
        double dx, dy;
        pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
        pref = mk((float)dx, (float)dy);
    }
    // sum of rewards: fixed-shape tree inside the wave, then per-arena in lane order
    if (p.actions && (p.flags & 2u)) {
        double r = active ? (double)rew : 0.0;
        const int w = P < 64 ? P : 64;
        for (int off = w >> 1; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
        if (active && (i & 63) == 0)
            atomicAdd(reinterpret_cast<double*>(&p.arena_stats[(size_t)a * ST_STRIDE + ST_SUMREW]), r);
    }
    // orientation of the observation frame (env.py:236): direction to the goal from the final state.
    // After an ORCA-only step or a reset `pref` already is that vector; otherwise derive it here, once
    // per agent, instead of in each of the 16 ray lanes of the observation kernel.
    float ox = pref.x, oy = pref.y;
    if (active && !do_reset && (p.actions != nullptr || goal_changed)) {
        double dx, dy;
        pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
        ox = (float)dx; oy = (float)dy;
    }
    CA_STAMP(10);
    __syncthreads();  // all lanes have read red[] and episode[]
    if (active) {
        p.orient_x[q] = ox; p.orient_y[q] = oy;
        p.pos_x[q] = pos.x; p.pos_y[q] = pos.y;
        p.vel_x[q] = vel.x; p.vel_y[q] = vel.y;
        p.pref_x[q] = pref.x; p.pref_y[q] = pref.y;
        if (i == 0) {
            unsigned long long* st = p.arena_stats + (size_t)a * ST_STRIDE;
            if (red[1]) st[ST_COLL] += (unsigned)red[1];
            if (red[2]) st[ST_OBST_COLL] += (unsigned)red[2];
            if (red[3]) st[ST_GOALS] += (unsigned)red[3];
            if (all_done) {  // + what a caller that auto-resets wants to know about the episode that ended
                st[ST_EPISODES] += 1;
                st[ST_LASTEP] = ((unsigned long long)(unsigned)steps << 32) | (unsigned)(N - red[0]);
            }
            p.arena_done[a] = all_done ? 1 : 0;
            p.step_count[a] = do_reset ? 0 : steps;
            if (do_reset) p.episode[a] = epi + 1;
        }
    }
    CA_STAMP(11);
}

// ============================================================================================
// reset() for every arena (env.py:461-488): new positions only; velocities, targets and the
// neighbour lists of the last step stay.
// ============================================================================================
__global__ void reset_kernel(const StepArgs p) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p.A * p.N) return;
    const int a = q / p.N, i = q - a * p.N;
    if (p.reset_mask && p.reset_mask[a] == 0) return;
    V2 pos;
    if (p.reset_px) {
        pos = mk(p.reset_px[q], p.reset_py[q]);
    } else {
        double u0, u1;
        rng2(p.seed, p.arena_offset + a, i, RNG_RESET, (uint32_t)p.episode[a], &u0, &u1);
        pos = mk((float)uniform64((double)p.spawn_x0, (double)p.spawn_x1, u0),
                 (float)uniform64((double)p.spawn_y0, (double)p.spawn_y1, u1));
    }
    double dx, dy;
    pref_dir64(pos.x, pos.y, p.goal_x[q], p.goal_y[q], &dx, &dy);
    p.pos_x[q] = pos.x; p.pos_y[q] = pos.y;
    p.pref_x[q] = (float)dx; p.pref_y[q] = (float)dy;
    p.orient_x[q] = (float)dx; p.orient_y[q] = (float)dy;
    p.agent_done[q] = 0;
}
// orientation from scratch (after the caller overwrote positions or goals through ca_set)
__global__ void orient_kernel(const StepArgs p) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p.A * p.N) return;
    double dx, dy;
    pref_dir64(p.pos_x[q], p.pos_y[q], p.goal_x[q], p.goal_y[q], &dx, &dy);
    p.orient_x[q] = (float)dx; p.orient_y[q] = (float)dy;
}
__global__ void reset_arena_kernel(const StepArgs p) {  // after reset_kernel: per-arena counters
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.A) return;
    if (p.reset_mask && p.reset_mask[a] == 0) return;
    p.step_count[a] = 0;
    p.arena_done[a] = 0;
    p.episode[a] += 1;
}

// ============================================================================================
// ALAN online action selection (ALAN_true.py:569-628), one lane per agent, around the ORCA step:
//   alan_select_kernel : softmax over the agent's action weights, one draw, preferred velocity =
//                        goal direction rotated by the chosen action (ALAN:578-598);
//   [nbr_kernel + step_kernel in ORCA mode: sim.doStep(), step counter, goal test (ALAN:601, 118-121)]
//   alan_update_kernel : reward of the executed action, sliding-window bandit update (ALAN:603-628).
// Weights, times and the reward that feeds them are fp64 like the reference's Python floats.
// ============================================================================================
enum { ALAN_MAX_ACTIONS = 32, ALAN_BS = 128 };
struct AlanArgs {
    const float *pos_x, *pos_y, *vel_x, *vel_y;
    const double *goal_x, *goal_y;
    float *pref_x, *pref_y, *reward;
    double *w, *t;        // [A*N][nA] action weights / time since the action's weight was set
    int* action;          // [A*N] the action of the current step (complemented while its arena sits out a step)
    double* dirs;         // [A*N][4] goal direction and rotated direction of the current step
    const double* u;      // [A*N] caller-supplied uniforms in [0,1), or null: Philox (RNG_ALAN, step)
    const int *step_count, *arena_done;
    unsigned long long* arena_stats;
    double act_c[ALAN_MAX_ACTIONS], act_s[ALAN_MAX_ACTIONS];  // (cos, sin) of every action's angle
    double temp, window, dt, reward_scale;
    uint64_t seed;
    int64_t arena_offset;
    int A, N, nA;
    uint32_t flags;
};

// numpy's float64 add.reduce for n < 128: < 8 sequential, otherwise eight accumulators combined as a
// fixed tree plus a sequential tail -- the value np.sum(ps) has at ALAN_true.py:582
template <class Get>
__device__ __forceinline__ double np_sum(int n, Get get) {
    if (n < 8) {
        double res = 0.0;
        for (int k = 0; k < n; ++k) res += get(k);
        return res;
    }
    double r0 = get(0), r1 = get(1), r2 = get(2), r3 = get(3), r4 = get(4), r5 = get(5), r6 = get(6), r7 = get(7);
    int k = 8;
    for (; k < n - (n % 8); k += 8) {
        r0 += get(k); r1 += get(k + 1); r2 += get(k + 2); r3 += get(k + 3);
        r4 += get(k + 4); r5 += get(k + 5); r6 += get(k + 6); r7 += get(k + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; k < n; ++k) res += get(k);
    return res;
}

__global__ __launch_bounds__(ALAN_BS) void alan_select_kernel(const AlanArgs p) {
    __shared__ double s_ps[ALAN_MAX_ACTIONS * ALAN_BS];  // [action][lane]
    const int q = blockIdx.x * ALAN_BS + threadIdx.x;
    if (q >= p.A * p.N) return;
    const int a = q / p.N, i = q - a * p.N, nA = p.nA;
    if ((p.flags & 16u) && p.arena_done[a] != 0) {  // CA_F_FREEZE: tell the update kernel, keep the last action
        p.action[q] = ~p.action[q];
        return;
    }
    double* ps = s_ps + threadIdx.x;
    const double* w = p.w + (size_t)q * nA;
    for (int k = 0; k < nA; ++k) ps[k * ALAN_BS] = exp64(w[k] / p.temp);           // ALAN:580-581
    const double sum = np_sum(nA, [&](int k) { return ps[k * ALAN_BS]; });
    double acc = 0.0;
    for (int k = 0; k < nA; ++k) {                                                 // ALAN:582
        const double v = ps[k * ALAN_BS] / sum;
        ps[k * ALAN_BS] = v;
        acc += v;
    }
    // np.random.choice(n, 1, p=ps) (ALAN:585): cdf = cumsum(p) / cdf[-1]; searchsorted(cdf, u, 'right')
    double ui;
    if (p.u) ui = p.u[q];
    else { double u1; rng2(p.seed, p.arena_offset + a, i, RNG_ALAN, (uint32_t)p.step_count[a], &ui, &u1); }
    int id = nA - 1;
    double run = 0.0;
    bool found = false;
    for (int k = 0; k < nA - 1; ++k) {
        run += ps[k * ALAN_BS];
        if (!found && run / acc > ui) { id = k; found = true; }
    }
    p.action[q] = id;
    double gx, gy;
    pref_dir64(p.pos_x[q], p.pos_y[q], p.goal_x[q], p.goal_y[q], &gx, &gy);        // ALAN:588
    const double cs = p.act_c[id], sn = p.act_s[id];                                // ALAN:592-595
    const double lx = gx * cs - gy * sn, ly = gx * sn + gy * cs;
    double* d = p.dirs + (size_t)q * 4;
    d[0] = gx; d[1] = gy; d[2] = lx; d[3] = ly;
    p.pref_x[q] = (float)lx; p.pref_y[q] = (float)ly;                               // ALAN:598
}

__global__ __launch_bounds__(ALAN_BS) void alan_update_kernel(const AlanArgs p) {
    const int q = blockIdx.x * ALAN_BS + threadIdx.x;
    if (q >= p.A * p.N) return;
    const int id = p.action[q];
    if (id < 0) {  // the arena was frozen when this step began
        p.action[q] = ~id;
        return;
    }
    const int a = q / p.N, nA = p.nA;
    const double* d = p.dirs + (size_t)q * 4;
    const float vxf = p.vel_x[q], vyf = p.vel_y[q];
    {   // env.py:389-400 in fp32, as ca_step reports it
        const float scale = (float)p.reward_scale;
        const float r_goal = vxf * (float)d[0] + vyf * (float)d[1];
        const float r_polite = vxf * (float)d[2] + vyf * (float)d[3];
        const float rew = scale * r_goal + (1.0f - scale) * r_polite;
        p.reward[q] = rew;
        if (p.flags & 2u)
            atomicAdd(reinterpret_cast<double*>(&p.arena_stats[(size_t)a * ST_STRIDE + ST_SUMREW]), (double)rew);
    }
    const double vx = (double)vxf, vy = (double)vyf;                                // ALAN:606-613
    const double R = p.reward_scale * (vx * d[0] + vy * d[1]) + (1.0 - p.reward_scale) * (vx * d[2] + vy * d[3]);
    double* w = p.w + (size_t)q * nA;
    double* t = p.t + (size_t)q * nA;
    for (int k = 0; k < nA; ++k) {                                                  // ALAN:616-628
        double tk = t[k] + p.dt;
        double wk = w[k];
        if (tk >= p.window) { tk = 0.0; wk = 0.0; }
        if (k == id) wk = R;
        t[k] = tk; w[k] = wk;
    }
    // the solve kernel left the goal direction in pref (its ORCA-mode epilogue); the reference's agent
    // still holds the velocity it was given at ALAN:598
    p.pref_x[q] = (float)d[2]; p.pref_y[q] = (float)d[3];
}

// ============================================================================================
// Laser observation (env.py:231-318, utils.py:5-113).
//
// A workgroup (256 lanes) owns 16 agents of ONE arena; 16 lanes per agent.  The arena's
// positions/velocities are staged in LDS once (the neighbour gathers then never leave the CU).
// Phase A -- lane per (source, ray) pair: the 16 lanes of an agent walk the pairs (8 octagon chords
//   per ORCA agent neighbour, one segment per ORCA obstacle neighbour), rotate each segment into
//   the goal-aligned frame and test it only against the rays that can possibly reach it: the
//   rays inside the segment's angular span as seen from the origin (a conservative superset, see
//   ray_span).  The ray/segment test itself is the reference's arithmetic, so culling never
//   changes a result.  A hit is merged into the ray's slot with one LDS ds_min_u64 on the key
//   (distance bits << 32 | segment index): the minimum distance wins and equal distances resolve
//   to the first segment, exactly like a serial first-minimum scan.
// Phase B -- lane per RAY: re-derives the winning segment's hit point and velocity and writes its
//   4 floats; the 16 lanes of an agent write its 256-B row, a wave stores 1 KiB contiguously.
// ============================================================================================
struct ObsArgs {
    const float *pos_x, *pos_y, *vel_x, *vel_y, *orient_x, *orient_y;
    const int *nb_count, *nb_idx, *obst_count, *obst_idx;
    const ObstDev* obst;
    float* obs;
    int A, N, K, S, bpa;  // bpa = workgroups per arena = ceil(N / 16)
    int a0;               // first arena of this launch
    unsigned long long* dbg;  // CA_STAMPS diagnostic build only: [waves][16] phase time stamps
    float radius;         // of the octagon = agent radius (env.py:31,338)
    float rays[32];       // env.py:321-332
    float oct[32];        // env.py:335-350
};

#ifdef CA_STAMPS
#define CA_OSTAMP(k)                                                                       \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        const unsigned long long _t = CA_STAMP_CLOCK();                                    \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                \
        if ((threadIdx.x & 63) == 0 && p.dbg)                                              \
            p.dbg[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (k)] = _t; \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
#else
#define CA_OSTAMP(k) do { } while (0)
#endif
// The observation workgroup: OBS_BS lanes = OBS_BS/16 agents of ONE arena (template parameter: 256,
// 512 or 1024 lanes, so that a workgroup can own a whole arena of up to 64 agents and stage it once).
constexpr int OBS_PAIRCAP = 16 * (16 + 8);  // (source, ray) pairs of one agent: <= 16 rays x (K + S) sources

// LDS (bytes): arena px,py,vx,vy [N] | keys [16][16] u64 | hit points [16][16] float2 | agent frames [16] float4 | nb idx [16][16] | obstacle idx [16][8]
//              | ray and octagon tables [64] | pair counts [2][16] | (source, ray) pair lists [16][384] u16
//              (a list holds the agent-neighbour pairs from its front and the obstacle pairs from its back)
__host__ __device__ inline size_t obs_lds_bytes(int N, int obs_bs) {
    const size_t apb = obs_bs / 16;
    return (size_t)N * 16 + 2 * apb * 16 * 8 + apb * 16 + apb * 16 * 4 + apb * 8 * 4 + 64 * 4 + 2 * apb * 4 + apb * OBS_PAIRCAP * 2;
}
#ifndef CA_OBS_BS_MAX
#define CA_OBS_BS_MAX 256
#endif
// 256 lanes measured best (C3: 99 us; 512 lanes: see profiles/r01_k_obs_variants.txt; 1024 lanes: 127 us)
__host__ __device__ inline int obs_block_threads(int N) {
    const int want = N > 32 ? 1024 : (N > 16 ? 512 : 256);
    return want < CA_OBS_BS_MAX ? want : CA_OBS_BS_MAX;
}

struct SegGeom {  // one segment in the goal-aligned frame, in the reference's intermediate terms
    float s02x, s02y;  // utils.py:19-20  p0 - p2, p0 = (0,0)
    float s32x, s32y;  // utils.py:11-12  p3 - p2
    float t_numer;     // utils.py:26
    float r1x, r1y, r2x, r2y;
};

// position of a point on the "ray dial": ray i points along (cos(i d), -sin(i d)), d = 2pi/16;
// returns u in [0,16) with |error| < 1e-4.  Only used to pick candidate rays (never for results).
__device__ __forceinline__ float ray_dial(float x, float y) {
    const float yy = -y;
    const float ax = fabsf(x), ay = fabsf(yy);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(mx);
    const float s = a * a;
    float r = a * (0.99997726f + s * (-0.33262347f + s * (0.19354346f + s * (-0.11643287f +
              s * (0.05265332f + s * -0.01172120f)))));
    r = (ay > ax) ? 1.57079632679f - r : r;
    r = (x < 0.0f) ? 3.14159265359f - r : r;
    r = (yy < 0.0f) ? -r : r;
    const float u = r * 2.54647908947f;  // 16 / (2 pi)
    return (u < 0.0f) ? u + 16.0f : u;
}

// Ray windows.  For a SEGMENT with end points p2, p3 every ray the exact test could accept lies in
// the angular span between the directions of p2 and p3 (short way round): the test accepts a ray
// only if its direction is between them up to fp32 rounding of two cross products, i.e. up to
// ~1e-7 rad unless an end point is very close to the origin compared with the other; the dial error
// is < 1e-4 and the margin is 0.01 dial units (3.9e-3 rad).  Segments passing (almost) through the
// origin, where "short way round" is ill-defined, get all 16 rays.  For an agent NEIGHBOUR the
// window is that of the circle through its octagon's vertices (see the pre-pass).
template <int OBS_BS>
__global__ __launch_bounds__(OBS_BS) void obs_kernel(const ObsArgs p) {
    constexpr int OBS_APB = OBS_BS / 16;  // agents per workgroup
    extern __shared__ float4 smem4[];
    const int tid = threadIdx.x;
    const int g = tid >> 4, r = tid & 15;
    const int N = p.N, K = p.K, S = p.S;
    const int ab = blockIdx.x / p.bpa;
    const int a = p.a0 + ab;
    const int i = (blockIdx.x - ab * p.bpa) * OBS_APB + g;
    const bool active = i < N;
    const size_t q = (size_t)a * N + (active ? i : 0);

    float* s_px = reinterpret_cast<float*>(smem4);
    float* s_py = s_px + N;
    float* s_vx = s_py + N;
    float* s_vy = s_vx + N;
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_vy + N);  // N*16 B: 8-aligned
    float2* s_hit = reinterpret_cast<float2*>(s_key + OBS_APB * 16);             // hit point of the key's chord
    float4* s_frame = reinterpret_cast<float4*>(s_hit + OBS_APB * 16);           // (cos, sin, pos x, pos y) per agent
    int* s_nb = reinterpret_cast<int*>(s_frame + OBS_APB);
    int* s_ob = s_nb + OBS_APB * 16;
    float* s_rays = reinterpret_cast<float*>(s_ob + OBS_APB * 8);  // [32] rays then [32] octagon
    float* s_oct = s_rays + 32;
    int* s_cnt = reinterpret_cast<int*>(s_oct + 32);                       // [16] neighbour pairs per agent
    int* s_cnt2 = s_cnt + OBS_APB;                                         // [16] obstacle pairs per agent
    unsigned short* s_pair = reinterpret_cast<unsigned short*>(s_cnt2 + OBS_APB);  // [16][OBS_PAIRCAP]
    CA_OSTAMP(0);
    if (tid < 32) { s_rays[tid] = p.rays[tid]; s_oct[tid] = p.oct[tid]; }

    for (int t = tid; t < N; t += OBS_BS) {
        const size_t qa = (size_t)a * N + t;
        s_px[t] = p.pos_x[qa]; s_py[t] = p.pos_y[qa]; s_vx[t] = p.vel_x[qa]; s_vy[t] = p.vel_y[qa];
    }
    int nn = 0, ns = 0;
    float c = 1.0f, s = 0.0f;
    if (active) {
        nn = p.nb_count[q]; ns = p.obst_count[q];
        c = p.orient_x[q]; s = -p.orient_y[q];  // utils.py:48-51: cos/sin of -atan2(orientation)
        if (r < nn) s_nb[g * 16 + r] = p.nb_idx[((size_t)a * K + r) * N + i];
        if (r < ns) s_ob[g * 8 + r] = p.obst_idx[((size_t)a * S + r) * N + i];
    }
    s_key[g * 16 + r] = ~0ull;
    if (r == 0) { s_cnt[g] = 0; s_cnt2[g] = 0; }
    CA_OSTAMP(1);
    __syncthreads();
    CA_OSTAMP(2);

    const int M = 8 * nn + ns;
    float mx = 0.0f, my = 0.0f;
    if (M > 0) { mx = s_px[i]; my = s_py[i]; }
    if (r == 0) s_frame[g] = make_float4(c, s, mx, my);  // phase A lanes also work for the wave's other agents
    // ---- pre-pass: which (source, ray) pairs are worth the exact test?  Supersets only; never results. ----
    // (1) lane per agent NEIGHBOUR: all 8 octagon vertices lie on the circle of radius R around it, so the
    // rays within asin(R/d) of its direction are a superset for each of its 8 chords.  asin(t) <= t + (pi/2 - 1) t^3
    // on [0, 1] ((asin t - t) / t^3 grows from 1/6 to pi/2 - 1); margin 0.02 dial units = 7.8e-3 rad covers the
    // dial's 1e-4 and the approximate reciprocal square root.
    for (int k = r; k < nn; k += 16) {
        const int nb = s_nb[g * 16 + k];
        const float rx = s_px[nb] - mx, ry = s_py[nb] - my;
        const float d2 = rx * rx + ry * ry, R = p.radius;
        const float ax = c * rx - s * ry, ay = s * rx + c * ry;
        const bool all = !(d2 > 1.0404f * R * R);  // the agent is inside (or within 2 % of) that circle
        const float ua = ray_dial(ax, ay);
        const float t = R * __builtin_amdgcn_rsqf(d2);
        const float hw = t * (1.0f + 0.5708f * (t * t)) * 2.54647908947f + 0.02f;
        int i0 = (int)ceilf(ua - hw), i1 = (int)floorf(ua + hw);
        if (all || i1 - i0 >= 15) { i0 = 0; i1 = 15; }
        const int w = i1 - i0 + 1;
        if (w > 0) {  // neighbour pairs fill the list from the front (list order is irrelevant: commutative minimum)
            const int base = atomicAdd(&s_cnt[g], w);
            for (int t2 = 0; t2 < w; ++t2)
                s_pair[g * OBS_PAIRCAP + base + t2] = (unsigned short)((k << 4) | ((i0 + t2) & 15));
        }
    }
    // (2) lane per RAY, one obstacle edge at a time: the exact test can accept a ray only if the ray's line
    // separates the edge's end points and the crossing is not behind the origin, i.e. (up to rounding, covered
    // by tolE = 50x the error of these products) the two end points are not on the same side of the line and
    // not both behind.  Obstacle pairs fill the agent's list from the back.
    {
        const float dx = s_rays[2 * r], dy = s_rays[2 * r + 1];
        int cnt2 = 0;
        for (int sidx = 0; sidx < ns; ++sidx) {
            const ObstDev o1 = load_obst(p.obst, s_ob[g * 8 + sidx]);
            const float x1 = o1.px - mx, y1 = o1.py - my, x2 = o1.qx - mx, y2 = o1.qy - my;
            const float ax = c * x1 - s * y1, ay = s * x1 + c * y1;
            const float bx = c * x2 - s * y2, by = s * x2 + c * y2;
            const float c2 = dx * ay - dy * ax, c3 = dx * by - dy * bx;
            const float f2 = dx * ax + dy * ay, f3 = dx * bx + dy * by;
            const float tolE = 1e-5f * p.rays[0] * (fabsf(ax) + fabsf(ay) + fabsf(bx) + fabsf(by) + 1.0f);
            const bool keep = !(c2 > tolE && c3 > tolE) && !(c2 < -tolE && c3 < -tolE) && (fmaxf(f2, f3) >= -tolE);
            const unsigned grp = (unsigned)(__ballot(keep) >> (threadIdx.x & 48)) & 0xFFFFu;  // my agent's 16 lanes
            if (keep)
                s_pair[g * OBS_PAIRCAP + OBS_PAIRCAP - 1 - (cnt2 + __popc(grp & ((1u << r) - 1u)))] =
                    (unsigned short)(((nn + sidx) << 4) | r);
            cnt2 += __popc(grp);
        }
        if (r == 0) s_cnt2[g] = cnt2;
    }
    CA_OSTAMP(3);
    wave_lds_sync();  // the 16 lanes of an agent are in one wave: no workgroup barrier needed
    CA_OSTAMP(4);
    // segment m of this agent in the rotated frame (env.py:283-294, 305-315; utils.py:55-62)
    auto build = [&](int m, SegGeom& sg, float& velx, float& vely, bool want_vel) {
        float x1, y1, x2, y2, vx = 0.0f, vy = 0.0f;
        if (m < 8 * nn) {
            const int k = m >> 3, e = m & 7;
            const int nb = s_nb[g * 16 + k];
            const float rx = s_px[nb] - mx, ry = s_py[nb] - my;
            const float4 oc = reinterpret_cast<const float4*>(s_oct)[e];
            x1 = oc.x + rx; y1 = oc.y + ry;
            x2 = oc.z + rx; y2 = oc.w + ry;
            if (want_vel) { vx = s_vx[nb]; vy = s_vy[nb]; }  // env.py:252
        } else {
            const ObstDev o1 = load_obst(p.obst, s_ob[g * 8 + (m - 8 * nn)]);
            x1 = o1.px - mx; y1 = o1.py - my;
            x2 = o1.qx - mx; y2 = o1.qy - my;
        }
        sg.r1x = c * x1 - s * y1; sg.r1y = s * x1 + c * y1;  // utils.py:59
        sg.r2x = c * x2 - s * y2; sg.r2y = s * x2 + c * y2;  // utils.py:60
        sg.s32x = sg.r2x - sg.r1x; sg.s32y = sg.r2y - sg.r1y;
        sg.s02x = 0.0f - sg.r1x; sg.s02y = 0.0f - sg.r1y;
        sg.t_numer = sg.s32x * sg.s02y - sg.s32y * sg.s02x;
        if (want_vel) {
            const float lvx = x1 + vx, lvy = y1 + vy;                      // utils.py:57
            const float rvx = c * lvx - s * lvy, rvy = s * lvx + c * lvy;  // utils.py:61
            velx = rvx - sg.r1x; vely = rvy - sg.r1y;                      // utils.py:62
        }
    };
    // chord e of neighbour slot k of agent ga (one of this wave's four), in ga's frame fr = (cos, sin, x, y)
    auto build_nb = [&](int ga, const float4& fr, int k, int e, SegGeom& sg) {
        const int nb = s_nb[ga * 16 + k];
        const float rx = s_px[nb] - fr.z, ry = s_py[nb] - fr.w;
        const float4 oc = reinterpret_cast<const float4*>(s_oct)[e];
        const float x1 = oc.x + rx, y1 = oc.y + ry, x2 = oc.z + rx, y2 = oc.w + ry;
        sg.r1x = fr.x * x1 - fr.y * y1; sg.r1y = fr.y * x1 + fr.x * y1;  // utils.py:59
        sg.r2x = fr.x * x2 - fr.y * y2; sg.r2y = fr.y * x2 + fr.x * y2;  // utils.py:60
        sg.s32x = sg.r2x - sg.r1x; sg.s32y = sg.r2y - sg.r1y;
        sg.s02x = 0.0f - sg.r1x; sg.s02y = 0.0f - sg.r1y;
        sg.t_numer = sg.s32x * sg.s02y - sg.s32y * sg.s02x;
    };
    // utils.py:5-40 for the ray with end point (s10x, s10y) starting at the origin.  (Deferring the
    // division/sqrt/atomic of accepted pairs to a second loop over a hit bitmask, and a branch-free
    // accept test, were both measured SLOWER: 136-138 us vs 117 us at C3.)
    auto hit = [&](const SegGeom& sg, float s10x, float s10y, float& d, float& hx, float& hy) -> bool {
        const float denom = s10x * sg.s32y - sg.s32x * s10y;          // utils.py:14
        if (denom == 0.0f) return false;
        const bool dpos = denom > 0.0f;
        const float s_numer = s10x * sg.s02y - s10y * sg.s02x;        // utils.py:21
        if ((s_numer < 0.0f) == dpos) return false;
        if ((sg.t_numer < 0.0f) == dpos) return false;
        if (((s_numer > denom) == dpos) || ((sg.t_numer > denom) == dpos)) return false;
        const float t = sg.t_numer / denom;                            // utils.py:34
        hx = 0.0f + t * s10x; hy = 0.0f + t * s10y;                    // utils.py:36-37
        d = sqrtf(hx * hx + hy * hy);                                  // utils.py:38
        return true;
    };

    // the same test and distance without early exits (the division of a rejected chord is computed and dropped)
    auto hit_nb = [&](const SegGeom& sg, float s10x, float s10y, float& d, float& hx, float& hy) -> bool {
        const float denom = s10x * sg.s32y - sg.s32x * s10y;          // utils.py:14
        const float s_numer = s10x * sg.s02y - s10y * sg.s02x;        // utils.py:21
        const bool dpos = denom > 0.0f;
        const bool ok = (denom != 0.0f) && ((s_numer < 0.0f) != dpos) && ((sg.t_numer < 0.0f) != dpos) &&
                        ((s_numer > denom) != dpos) && ((sg.t_numer > denom) != dpos);  // utils.py:15-31
        const float t = sg.t_numer / denom;                            // utils.py:34
        hx = 0.0f + t * s10x; hy = 0.0f + t * s10y;                    // utils.py:36-37
        d = sqrtf(hx * hx + hy * hy);                                  // utils.py:38
        return ok;
    };

    // ---- phase A: lane per (source, ray) pair ----
    // An agent neighbour contributes the 8 chords of its octagon; consecutive chords share an end point
    // bit for bit (env.py:335-350 builds them as a chain), so the 8 rotated vertices are computed once
    // per pair and every chord is accept-tested against the pair's single ray.  Only accepted chords
    // (about two per pair) are re-derived through build()/hit() for the exact hit distance.
    const float tol = 2e-5f * p.rays[0] * (p.rays[0] + 2.0f * p.radius + 1.0f);  // rays[0] = neighbor_dist (env.py:321-332)
    // The lane whose key is the ray's minimum after this trip's atomics leaves its hit point next to the key
    // (the LDS executes one wave's instructions in order, so the re-read sees every lane's atomic of the trip;
    // a later, smaller key overwrites both).  Phase B then needs no second division / square root.
    auto merge = [&](int ga, int ray, float best, int best_m, float bhx, float bhy) {
        if (best_m >= 0) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)best_m;
            atomicMin(&s_key[ga * 16 + ray], key);
            if (s_key[ga * 16 + ray] == key) s_hit[ga * 16 + ray] = make_float2(bhx, bhy);
        }
    };
    // (neighbour, ray) pairs and (obstacle edge, ray) pairs in loops of their own: a wave that mixes the two
    // kinds in one pass pays for both code paths.  The neighbour pairs of the wave's FOUR agents form one
    // work list shared by its 64 lanes (an agent has 13 pairs on average but often a few more than 16, which
    // would cost its 16 lanes -- and with them the wave -- a second trip).
    const int g0 = g & ~3;
    const int n0 = s_cnt[g0], n1 = s_cnt[g0 + 1], n2 = s_cnt[g0 + 2], n3 = s_cnt[g0 + 3];
    const int ntot = n0 + n1 + n2 + n3;
    for (int pi = tid & 63; pi < ntot; pi += 64) {
        int ga = g0, li = pi;
        { const bool b = li >= n0; ga = b ? g0 + 1 : ga; li = b ? li - n0 : li;
          const bool b1 = b && li >= n1; ga = b1 ? g0 + 2 : ga; li = b1 ? li - n1 : li;
          const bool b2 = b1 && li >= n2; ga = b2 ? g0 + 3 : ga; li = b2 ? li - n2 : li; }
        const float4 fr = s_frame[ga];
        const int pr = s_pair[ga * OBS_PAIRCAP + li];
        const int k = pr >> 4, ray = pr & 15;
        const float s10x = s_rays[2 * ray] - 0.0f, s10y = s_rays[2 * ray + 1] - 0.0f;
        float best = __int_as_float(0x7f800000), bhx = 0.0f, bhy = 0.0f;
        int best_m = -1;
        const int nb = s_nb[ga * 16 + k];
        const float rx = s_px[nb] - fr.z, ry = s_py[nb] - fr.w;
        // Which chords can the exact test accept?  It needs the crossing parameter along the chord,
        // s_numer / denom (utils.py:21-31), inside [0, 1], i.e. the ray's LINE must separate the chord's end
        // points: with cr[e] = ray x vertex e (= -s_numer of chord e), a chord whose two end points lie on the
        // same side of the line by more than `tol` cannot be accepted.  The filter does not need the reference's
        // rounding, so it takes the cross products in the WORLD frame, where the octagon's vertices are constants:
        // ray_w x (oct_e + rel) with ray_w the ray turned back by the agent's frame -- no vertex is rotated here.
        // `tol` is 100x the rounding error of either form.  The survivors -- the entry and the exit chord, a
        // third one when the line grazes a vertex -- go through the reference's arithmetic below.
        const float wx = fr.x * s10x + fr.y * s10y, wy = fr.x * s10y - fr.y * s10x;
        const float wb = wx * ry - wy * rx;
        float cr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 oc = reinterpret_cast<const float4*>(s_oct)[e];
            cr[e] = (wx * oc.y - wy * oc.x) + wb;
        }
        unsigned acc = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ca = cr[e], cb = cr[(e + 1) & 7];
            const bool same_side = (ca > tol && cb > tol) || (ca < -tol && cb < -tol);
            acc |= same_side ? 0u : (1u << e);
        }
        // ascending chord index, strict '<': the first minimum wins.  Two surviving chords (entry and exit)
        // are evaluated side by side in straight-line code, so that the two instruction streams can share
        // packed fp32 instructions; the arithmetic of each is the reference's (utils.py:14-38).
        while (acc) {
            const int e1 = __ffs(acc) - 1;
            acc &= acc - 1;
            const bool two = acc != 0;
            const int e2 = two ? __ffs(acc) - 1 : e1;
            acc &= acc - 1;  // (0 & anything stays 0)
            SegGeom g1, g2;
            build_nb(ga, fr, k, e1, g1);
            build_nb(ga, fr, k, e2, g2);
            float d1, d2, h1x, h1y, h2x, h2y;
            const bool ok1 = hit_nb(g1, s10x, s10y, d1, h1x, h1y), ok2 = hit_nb(g2, s10x, s10y, d2, h2x, h2y) && two;
            if (ok1 && d1 < best) { best = d1; best_m = 8 * k + e1; bhx = h1x; bhy = h1y; }
            if (ok2 && d2 < best) { best = d2; best_m = 8 * k + e2; bhx = h2x; bhy = h2y; }
        }
        merge(ga, ray, best, best_m, bhx, bhy);
    }
    const int no = s_cnt2[g];
    for (int pi = r; pi < no; pi += 16) {
        const int pr = s_pair[g * OBS_PAIRCAP + OBS_PAIRCAP - 1 - pi];
        const int k = pr >> 4, ray = pr & 15;
        const float s10x = s_rays[2 * ray] - 0.0f, s10y = s_rays[2 * ray + 1] - 0.0f;
        SegGeom sg;
        float dum0, dum1, d, hx, hy;
        const int m = 8 * nn + (k - nn);
        build(m, sg, dum0, dum1, false);
        if (hit(sg, s10x, s10y, d, hx, hy)) merge(g, ray, d, m, hx, hy);
    }
    CA_OSTAMP(5);
    wave_lds_sync();
    CA_OSTAMP(6);
    if (!active) return;
    // ---- phase B: lane per ray ----
    const unsigned long long key = s_key[g * 16 + r];
    float bx = 0.0f, by = 0.0f, vx = 0.0f, vy = 0.0f;
    if (key != ~0ull) {
        SegGeom sg;
        float wx, wy;
        build((int)(unsigned)key, sg, wx, wy, true);
        const float2 h = s_hit[g * 16 + r];
        bx = h.x; by = h.y;
        if (!(bx == 0.0f && by == 0.0f)) { vx = wx; vy = wy; }  // utils.py:103
    }
    CA_OSTAMP(7);
    reinterpret_cast<float4*>(p.obs)[q * 16 + r] = make_float4(bx, by, vx, vy);
    CA_OSTAMP(8);
}

// ---- diagnostics for the numerics contract ----
__global__ void debug_math_kernel(int op, const void* in, void* out, int n, uint64_t seed) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (op == 0) {
        ((float*)out)[t] = sqrtf(((const float*)in)[t]);
    } else if (op == 1) {
        ((float*)out)[t] = ((const float*)in)[2 * t] / ((const float*)in)[2 * t + 1];
    } else if (op == 2) {
        double s, c;
        sincos64(((const double*)in)[t], &s, &c);
        ((double*)out)[2 * t] = s; ((double*)out)[2 * t + 1] = c;
    } else if (op == 3) {
        const float* f = (const float*)in + 4 * t;
        double x, y;
        pref_dir64(f[0], f[1], (double)f[2], (double)f[3], &x, &y);
        ((double*)out)[2 * t] = x; ((double*)out)[2 * t + 1] = y;
    } else if (op == 4) {
        const uint32_t* u = (const uint32_t*)in + 4 * t;
        double a, b;
        rng2(seed, (int64_t)u[0], (int)u[1], (int)u[2], u[3], &a, &b);
        ((double*)out)[2 * t] = a; ((double*)out)[2 * t + 1] = b;
    } else if (op == 5) {
        ((double*)out)[t] = exp64(((const double*)in)[t]);
    }
}

}  // namespace ca
