// ca_lp.h -- 2-D linear programs of ORCA (App. A.5): LP1, LP2, LP3 over LDS, private and register line tables
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
#pragma once
#include "ca_common.h"

namespace ca {

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
    const float sq = sqrt_ir(disc);   // (disc = 0 or >= an ulp of O(1) terms; negative: the lane has failed already / fails below)
    float tLeft = -dp - sq;
    float tRight = -dp + sq;
    bool failed = false;
    auto clip = [&](const Line& M) {
        const float den = det(L.dir, M.dir);
        const float num = det(M.dir, L.point - M.point);
        const bool par = fabsf(den) <= EPS;
        const float t = div_ir(num, den);   // (used only where |den| > EPS: den in (1e-5, 1], num = 0 or in [1e-21, 5e7])
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
    else if (absSq(opt) > sqr(radius)) result = normalize_ir(opt) * radius;   // (|opt| > radius here)
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
__device__ __noinline__ void lp3(__attribute__((address_space(3))) char* lines3, int stride, int n, int numObst, int begin, float radius,
                                V2& result) {
    LdsLines ls; ls.base = (float4*)lines3; ls.stride = stride;   // (the table is LDS and crosses the call boundary as such: no FLAT accesses)
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
                    l.point = Li.point + div_ir(det(Lj.dir, Li.point - Lj.point), d) * Li.dir;   // (|d| > EPS here)
                }
                l.dir = normalize_ir(Lj.dir - Li.dir);   // (unit vectors that are not parallel: |difference| in [1e-5, 2]; parallel ones are skipped or opposite)
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

// (opt_fn() yields the optimisation point: a register pair, or -- for the workgroup shapes that run out of registers --
// a read of the lane's LDS slot at each use, so that the value does not live in registers across the whole solve)
template <int ML, int ST, int I, class OptFn>
__device__ __forceinline__ bool lp1_reg(const float4 (&L)[ML], int no, float radius, OptFn opt_fn, V2& result) {
    const Line Li = unpack_line(L[I]);
    const float dp = dot(Li.point, Li.dir);
    const float disc = sqr(dp) + sqr(radius) - absSq(Li.point);
    if (disc < 0.0f) return false;
    const float sq = sqrt_ir(disc);   // (disc = 0 or >= an ulp of O(1) terms; negative: the lane has failed already / fails below)
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
            const float t = div_ir(num, den);   // (used only where |den| > EPS: den in (1e-5, 1], num = 0 or in [1e-21, 5e7])
            // (bitwise operators on purpose: with && / || the compiler builds branches around single moves -- 546 branches and
            // 4 754 scalar instructions in the K = 10 kernel against 374 and 3 407 this way)
            const bool pos = den >= 0.0f;
            const bool right = !par & pos, left = !par & !pos;
            tRight = (right & (t < tRight)) ? t : tRight;
            tLeft = (left & (tLeft < t)) ? t : tLeft;
            failed |= (par & (num < 0.0f)) | (!par & (tLeft > tRight));
        }
    });
    if (failed) return false;
    const V2 opt = opt_fn();
    const float t = dot(Li.dir, opt - Li.point);
    if (t < tLeft) result = Li.point + tLeft * Li.dir;
    else if (t > tRight) result = Li.point + tRight * Li.dir;
    else result = Li.point + t * Li.dir;
    return true;
}

// App. A.5 LP2 (dirOpt = false) over the register slots; returns the contract's line index of the
// first infeasible line, or the line count when all lines are satisfied.
template <int ML, int ST, class OptFn>
__device__ __forceinline__ int lp2_reg(const float4 (&L)[ML], int no, int ncnt, float radius, OptFn opt_fn, V2& result) {
    {
        const V2 opt = opt_fn();
        if (absSq(opt) > sqr(radius)) result = normalize_ir(opt) * radius;   // (|opt| > radius here)
        else result = opt;
    }
    int fail = no + ncnt;
    bool alive = true;
    static_for<ML>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const bool valid = (i < ST) ? (i < no) : (i - ST < ncnt);
        if constexpr (i == ST + (ML - ST) / 2) CA_PRIO_POINT(4);
        if (alive && valid) {
            const Line Li = unpack_line(L[i]);
            if (det(Li.dir, Li.point - result) > 0.0f) {
                const V2 tmp = result;
                if (!lp1_reg<ML, ST, i>(L, no, radius, opt_fn, result)) {
                    result = tmp;
                    fail = (i < ST) ? i : no + (i - ST);
                    alive = false;
                }
            }
        }
    });
    return fail;
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
// value of the lane whose index differs in bit 0 (CTRL = 0xB1: quad_perm [1,0,3,2]) or bit 1 (0x4E: [2,3,0,1])
template <int CTRL>
__device__ __forceinline__ int quad_xor(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ float quad_xor(float v) { return __int_as_float(quad_xor<CTRL>(__float_as_int(v))); }

// (stride: slots per row of the pool -- POOL_SLOTS; 1 for the single-agent table of ca_step.h solve_many_obstacles)
// (Round 5 built two variants in which a group goes straight to ITS next violated line, so that the wave runs the long body
// max-over-groups times instead of once per position of the union -- each group walking there one read at a time: 4-8 % slower
// everywhere; the group's four lanes testing the remaining lines together (two reads in flight, a quad minimum): C3 -0.7 %, C5
// +1.2 %, C2 +9 % -- neither kept: profiles/r05_b_simd_balance_and_lp3_walk.txt.)
// The pool is LDS, and lp3_coop is told so: through a generic `float4*` across the call boundary every access was a FLAT
// instruction (aperture check, both memory counters) in the middle of these latency chains; with an address-space-3 pointer they
// are ds_read_b128 / ds_write_b128.
typedef float lp3_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lp3_v4f lp3_lds_v4f;
struct Lds3Lines {  // [line][slot] float4 = (point.x, point.y, dir.x, dir.y), in LDS
    lp3_lds_v4f* base;  // already offset by the slot
    int stride;
    __device__ __forceinline__ Line get(int j) const {
        const lp3_v4f v = base[j * stride];
        Line l; l.point = mk(v.x, v.y); l.dir = mk(v.z, v.w);
        return l;
    }
    __device__ __forceinline__ void put(int j, const Line& l) const {
        const lp3_v4f v = {l.point.x, l.point.y, l.dir.x, l.dir.y};
        base[j * stride] = v;
    }
};
__device__ __noinline__ void lp3_coop_lds(lp3_lds_v4f* pool, int ML, int nslots, float radius, int stride) {
    const int lane = threadIdx.x & 63, slot = lane >> 2, q = lane & 3;
    lp3_lds_v4f* hdr = pool + (2 * ML - 1) * stride;
    Lds3Lines ls; ls.base = pool + slot; ls.stride = stride;
    Lds3Lines pj; pj.base = pool + ML * stride + slot; pj.stride = stride;
    const bool live = slot < nslots;
    lp3_v4f h = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) h = hdr[slot];
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
                        l.point = Li.point + div_ir(det(Lj.dir, Li.point - Lj.point), d) * Li.dir;   // (|d| > EPS here)
                    }
                    l.dir = normalize_ir(Lj.dir - Li.dir);   // (unit vectors that are not parallel: |difference| in [1e-5, 2]; parallel ones are skipped or opposite)
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
                    const float sq = sqrt_ir(disc);   // (disc = 0 or >= an ulp of O(1) terms; negative: the lane has failed already / fails below)
                    float tLeft = -dp - sq;
                    float tRight = -dp + sq;
                    for (int jj = q; jj < ii; jj += 4) {
                        const Line M = pj.get(jj);
                        const float den = det(L.dir, M.dir);
                        const float num = det(M.dir, L.point - M.point);
                        const bool par = fabsf(den) <= EPS;
                        const float t = div_ir(num, den);   // (used only where |den| > EPS: den in (1e-5, 1], num = 0 or in [1e-21, 5e7])
                        const bool right = !par && den >= 0.0f, left = !par && !(den >= 0.0f);
                        tRight = (right && t < tRight) ? t : tRight;
                        tLeft = (left && tLeft < t) ? t : tLeft;
                        failed |= (par && num < 0.0f) ? 1 : 0;
                    }
                    // merge over the group's four lanes: two quad-permute steps (lane ^ 1, then lane ^ 2) as DPP
                    // register moves -- a ds_bpermute per value and step would sit in this serial chain instead
                    {
                        const float oR = quad_xor<0xB1>(tRight), oL = quad_xor<0xB1>(tLeft);
                        tRight = (oR < tRight) ? oR : tRight;
                        tLeft = (tLeft < oL) ? oL : tLeft;
                        failed |= quad_xor<0xB1>(failed);
                    }
                    {
                        const float oR = quad_xor<0x4E>(tRight), oL = quad_xor<0x4E>(tLeft);
                        tRight = (oR < tRight) ? oR : tRight;
                        tLeft = (tLeft < oL) ? oL : tLeft;
                        failed |= quad_xor<0x4E>(failed);
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
    if (live && q == 0) { const lp3_v4f o = {result.x, result.y, h.z, 0.0f}; hdr[slot] = o; }
}
__device__ __forceinline__ void lp3_coop(float4* pool, int ML, int nslots, float radius, int stride = POOL_SLOTS) {
    lp3_coop_lds((lp3_lds_v4f*)pool, ML, nslots, radius, stride);
}

}  // namespace ca
