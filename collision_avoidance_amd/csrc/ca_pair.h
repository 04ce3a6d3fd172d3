// ca_pair.h -- the solve kernel with TWO LANES PER AGENT, for LARGE arenas (129 .. 512 agents per arena)
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
//
// Why: one lane per agent turns BASELINE config C5 (256 arenas x 512 agents) into 2048 waves on a chip of 1024 SIMDs:
// two waves per SIMD, each one long dependent chain (neighbour scan -> ten half-planes -> LP2 -> statistics) that runs
// as slowly alone as with three co-runners (a C5 wave takes as many cycles at two waves per SIMD as a C3 wave takes at
// four: profiles/r04_a_c5_phase_stamps_and_pair_kernel.txt) -- the chip idles half of its issue slots.  Here lanes 2 s and 2 s + 1 work for
// agent slot s, so the same arena is twice as many waves, each with a shorter chain:
//   * the arena is sorted into a uniform grid in LDS (counting sort, as in ca_nbr.h; cells a quarter of the neighbour range
//     wide) and FROM THEN ON the pair of slot s works for the agent at position s of the sorted list: a wave's 32 agents are
//     neighbours in space -- they walk the same cell rows (candidate reads are LDS broadcasts, one trip count per wave) and
//     need LP3 together or not at all;
//   * the scan covers only the cells within the distance of the FARTHEST MEMBER OF THE AGENT'S PREVIOUS LIST (any K distinct
//     agents bound the distance of the K-th nearest: a settled crowd has ~50 agents within the neighbour range of 5 and the
//     tenth-nearest at 2.6); every cell row's candidates are dealt to the two lanes alternately, each lane keeps a sorted list
//     of 64-bit (distance, index) keys and the two lists are merged by ONE quad-permute exchange and a bitonic network
//     (ca_quad.h merge_with_partner): the keys are totally ordered, so the merged list IS the serial scan's;
//   * lane h builds the half-planes of neighbours h, h + 2, ... into ITS register slots: slot m of the even lane holds
//     neighbour 2 m, slot m of the odd lane neighbour 2 m + 1 -- ONE instruction stream builds two lines at a time;
//   * LP2 walks the lines in the contract's order (the current line reaches both lanes by a DPP broadcast from its owner;
//     both lanes hold the same running result), and each LP1 inside it clips against the earlier lines two at a time:
//     "clip against my slot m" is lines 2 m and 2 m + 1 at once; tLeft (a maximum), tRight (a minimum) and the failure flag
//     (an OR) are merged over the pair -- all independent of the order, so the values are those of the serial loop bit for
//     bit (the argument of ca_lp.h lp3_coop / ca_quad.h lp2_quad, here on REGISTER lines);
//   * the obstacle half-planes (<= 4: the boundary polygon) are built by both lanes redundantly (their "already covered"
//     rule is sequential) and clipped against two at a time as well;
//   * the infeasible agents go through the per-wave LDS pool and lp3_coop like the lane kernel's;
//   * the per-agent scalar work (fp64 goal direction, done test, RNG) is done redundantly by the two lanes; lane 0 of a pair
//     writes.
// One workgroup = one arena = 2 P lanes (P = 256 or 512).  Selected by ca_create for arenas of 129 .. 512 agents with K <= 10 and
// <= 4 obstacle neighbours (CA_PAIR=0 falls back to the lane kernel, with helper lanes in its scan from 192 agents).  Measured
// crossover, settled crowds, ORCA-only us per step, lane / pair: 1024 x 100 (circle) 59.2 / 67.0, 1024 x 128 65.1 / 69.4 -- arenas
// of two waves stay on the lane kernel --, 512 x 180 71.7 / 49.2, 256 x 512 66.9 / 50.1.
#pragma once
#include "ca_quad.h"

namespace ca {

template <int OWNER>
__device__ __forceinline__ float pair_bcast(float v) {  // the value of lane OWNER (0 / 1) of the pair, in both lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), OWNER ? 0xF5 : 0xA0, 0xF, 0xF, false));  // quad_perm [1,1,3,3] / [0,0,2,2]
}
template <int OWNER>
__device__ __forceinline__ float4 pair_bcast4(const float4& v) {
    return make_float4(pair_bcast<OWNER>(v.x), pair_bcast<OWNER>(v.y), pair_bcast<OWNER>(v.z), pair_bcast<OWNER>(v.w));
}
__device__ __forceinline__ int pair_sum(int v) { return v + quad_xor<0xB1>(v); }

// One LP1 clip (App. A.5) of line Li against an earlier line M, branch-free (ca_lp.h lp1_reg), `valid` = this lane really
// holds such a line
struct Clip {
    float tLeft, tRight;
    bool failed;
    __device__ __forceinline__ void operator()(const Line& Li, const float4& Mv, bool valid) {
        const Line M = unpack_line(Mv);
        const float den = det(Li.dir, M.dir);
        const float num = det(M.dir, Li.point - M.point);
        const bool par = fabsf(den) <= EPS;
        const float t = div_ir(num, den);   // (ca_lp.h lp1_reg: in range wherever it is used)
        const bool pos = den >= 0.0f;
        const bool right = valid & !par & pos, left = valid & !par & !pos;
        tRight = (right & (t < tRight)) ? t : tRight;
        tLeft = (left & (tLeft < t)) ? t : tLeft;
        failed |= valid & par & (num < 0.0f);
    }
    // merge over the pair: the partner's interval and flag (order-free: max / min / or)
    __device__ __forceinline__ void merge() {
        const float oR = quad_xor<0xB1>(tRight), oL = quad_xor<0xB1>(tLeft);
        tRight = (oR < tRight) ? oR : tRight;
        tLeft = (tLeft < oL) ? oL : tLeft;
        failed |= quad_xor<0xB1>(failed ? 1 : 0) != 0;
    }
};

// App. A.5 LP2 (dirOpt = false), two lanes per agent, lines in registers:
//   OB[j]  obstacle line j, j < no (both lanes hold all of them);  OBP[t] = OB[2 t + h] (this lane's share of them in LP1)
//   LA[m]  the half-plane of neighbour 2 m + h (valid for 2 m + h < ncnt)
// Returns the contract's index of the first infeasible line, or no + ncnt.
template <int KMAX, class OptFn>
__device__ __forceinline__ int lp2_pair(const float4 (&OB)[4], const float4 (&OBP)[2], const float4 (&LA)[(KMAX + 1) / 2], int no, int ncnt,
                                        int h, float radius, OptFn opt_fn, V2& result) {
    {
        const V2 opt = opt_fn();
        if (absSq(opt) > sqr(radius)) result = normalize_ir(opt) * radius;   // (|opt| > radius here)
        else result = opt;
    }
    int fail = no + ncnt;
    bool alive = true;
    auto solve_on = [&](const Line& Li, Clip& c) __attribute__((always_inline)) -> bool {  // the end of LP1: both lanes alike
        c.merge();
        if (c.failed | (c.tLeft > c.tRight)) return false;
        const V2 opt = opt_fn();
        const float t = dot(Li.dir, opt - Li.point);
        if (t < c.tLeft) result = Li.point + c.tLeft * Li.dir;
        else if (t > c.tRight) result = Li.point + c.tRight * Li.dir;
        else result = Li.point + t * Li.dir;
        return true;
    };
    static_for<4>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pidx = decltype(pc)::value;
        if (alive && pidx < no) {
            const Line Li = unpack_line(OB[pidx]);
            if (det(Li.dir, Li.point - result) > 0.0f) {
                const V2 tmp = result;
                const float dp = dot(Li.point, Li.dir);
                const float disc = sqr(dp) + sqr(radius) - absSq(Li.point);
                bool ok = !(disc < 0.0f);
                if (ok) {
                    const float sq = sqrt_ir(disc);
                    Clip c; c.tLeft = -dp - sq; c.tRight = -dp + sq; c.failed = false;
                    if constexpr (pidx >= 1) c(Li, OBP[0], h < pidx);        // obstacle lines 0 / 1
                    if constexpr (pidx >= 3) c(Li, OBP[1], 2 + h < pidx);    // obstacle line 2 (3 is this one)
                    ok = solve_on(Li, c);
                }
                if (!ok) { result = tmp; fail = pidx; alive = false; }
            }
        }
    });
    static_for<KMAX>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (alive && k < ncnt) {
            const Line Li = unpack_line(pair_bcast4<(k & 1)>(LA[k >> 1]));  // from the lane that built it
            if (det(Li.dir, Li.point - result) > 0.0f) {
                const V2 tmp = result;
                const float dp = dot(Li.point, Li.dir);
                const float disc = sqr(dp) + sqr(radius) - absSq(Li.point);
                bool ok = !(disc < 0.0f);
                if (ok) {
                    const float sq = sqrt_ir(disc);
                    Clip c; c.tLeft = -dp - sq; c.tRight = -dp + sq; c.failed = false;
                    if (__builtin_amdgcn_ballot_w64(no > 0) != 0ull) {   // (interior waves have no obstacle line at all)
                        c(Li, OBP[0], h < no);
                        c(Li, OBP[1], 2 + h < no);
                    }
                    static_for<(k + 1) / 2>([&](auto mc) __attribute__((always_inline)) {
                        constexpr int m = decltype(mc)::value;
                        if constexpr (2 * m + 1 < k) c(Li, LA[m], true);     // lines 2 m and 2 m + 1, both earlier
                        else c(Li, LA[m], h == 0);                           // line k - 1 (even lane); the odd lane holds line k itself
                    });
                    ok = solve_on(Li, c);
                }
                if (!ok) { result = tmp; fail = no + k; alive = false; }
            }
        }
    });
    return fail;
}

// LDS of the pair kernel (bytes, dynamic): LP3 pool [waves = 2 BS / 64][2 ML][16] float4 | px py vx vy [BS] | misc [BS][4]
__host__ __device__ inline size_t pair_lds_bytes(int BS, int KMAX) {
    return (size_t)(2 * BS / 64) * (2 * (4 + KMAX)) * POOL_SLOTS * 16 + (size_t)BS * 32;
}

#ifdef CA_STAMPS
#define CA_PSTAMP(k)                                                                       \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        const unsigned long long _t = CA_STAMP_CLOCK();                                    \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                \
        if ((threadIdx.x & 63) == 0 && p.dbg)                                              \
            p.dbg[((size_t)blockIdx.x * (2 * BS / 64) + (threadIdx.x >> 6)) * 16 + (k)] = _t; \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
#else
#define CA_PSTAMP(k) do { } while (0)
#endif

// BS = agent slots of the workgroup = P (the arena's power-of-two size, 256 or 512); launched with 2 BS lanes
template <int KMAX, int BS>
__global__ __launch_bounds__(2 * BS, 4) void pair_kernel(const StepArgs p) {
    static_assert(BS == 256 || BS == 512, "one arena of 129 .. 512 agents per workgroup");
    constexpr int ST = 4, ML = ST + KMAX, KH = (KMAX + 1) / 2;
    constexpr int M = KMAX <= 4 ? 4 : (KMAX <= 8 ? 8 : 16);  // merge width of the neighbour lists
    constexpr int GMAX = 32;
    extern __shared__ float4 smem4[];
    __shared__ unsigned s_box[4];
    __shared__ int s_ccnt[GMAX * GMAX];
    __shared__ int s_cstart[GMAX * GMAX + 1];
    __shared__ unsigned short s_sorted[BS];
    __shared__ float2 s_sxy[BS];
    __shared__ int s_red[4];   // per-arena reductions of the epilogue
    __shared__ unsigned s_vmax2;   // the largest squared speed of the arena in this step, as float bits (the pair count's motion bound)
    const int tid = threadIdx.x;
    const int h = tid & 1, sl = tid >> 1;   // lane of the pair, agent slot of the pair
    const int a = p.a0 + (int)blockIdx.x;
    const int N = p.N, K = p.K, S = p.S;
    const bool frozen = arena_frozen(p, a);  // CA_F_FREEZE: the episode of this arena is over
    const bool in_arena = (a < p.a1) && (sl < N);
    const bool active = in_arena && !frozen;
    if (frozen && tid == 0) p.arena_stats[(size_t)a * ST_STRIDE + ST_FROZEN] += 1;

    float4* s_lines = smem4;  // [waves][2 ML][POOL_SLOTS] (last row: slot headers)
    float* s_px = reinterpret_cast<float*>(smem4 + (size_t)(2 * BS / 64) * (2 * ML) * POOL_SLOTS);
    float* s_py = s_px + BS;
    float* s_vx = s_py + BS;
    float* s_vy = s_vx + BS;
    int* s_misc = reinterpret_cast<int*>(s_vy + BS);  // [BS][4]: goal direction, preferred velocity (parked across the solve)

    CA_PSTAMP(0);
    // ---- stage the arena: slot sl loads agent sl (coalesced); preferred velocity of this step (env.py:371-383) ----
    int Gx, Gy;
    float gx0, gy0, gics;   // the grid: origin and 1 / cell size
    {
        V2 pos0 = mk(0.0f, 0.0f), vel0 = mk(0.0f, 0.0f), pref0 = mk(0.0f, 0.0f);
        V2 pf0 = mk(1.0f, 0.0f);
        if (active) {
            const int q0 = a * N + sl;
            pos0 = mk(p.pos_x[q0], p.pos_y[q0]);
            vel0 = mk(p.vel_x[q0], p.vel_y[q0]);
            if (p.actions) {
                double pf_x, pf_y, sn, cs64;
                pref_dir64(pos0.x, pos0.y, p.goal_x[q0], p.goal_y[q0], &pf_x, &pf_y);
                sincos64((double)p.actions[q0], &sn, &cs64);
                const double rl_x = pf_x * cs64 - pf_y * sn;
                const double rl_y = pf_x * sn + pf_y * cs64;
                pf0 = mk((float)pf_x, (float)pf_y);
                pref0 = mk((float)rl_x, (float)rl_y);
            } else {
                pref0 = mk(p.pref_x[q0], p.pref_y[q0]);
            }
        }
        if (h == 0) {
            s_px[sl] = pos0.x; s_py[sl] = pos0.y; s_vx[sl] = vel0.x; s_vy[sl] = vel0.y;
            reinterpret_cast<float*>(s_misc)[sl * 4 + 0] = pf0.x; reinterpret_cast<float*>(s_misc)[sl * 4 + 1] = pf0.y;
            reinterpret_cast<float*>(s_misc)[sl * 4 + 2] = pref0.x; reinterpret_cast<float*>(s_misc)[sl * 4 + 3] = pref0.y;
        }
        if (tid < 2) s_box[tid] = 0xFFFFFFFFu;
        if (tid >= 2 && tid < 4) s_box[tid] = 0u;
        if (tid >= 4 && tid < 8) s_red[tid - 4] = 0;
        if (tid == 8) s_vmax2 = 0u;
        for (int cidx = tid; cidx < GMAX * GMAX; cidx += 2 * BS) s_ccnt[cidx] = 0;
        __syncthreads();
        CA_PSTAMP(1);
        // ---- the uniform grid of ca_nbr.h: counting sort of the agents by cell (frozen arenas keep the barriers) ----
        auto ord = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        auto unord = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); };
        {
            const unsigned ox = ord(pos0.x), oy = ord(pos0.y);
            const unsigned bx0 = wave_min_u32(in_arena ? ox : 0xFFFFFFFFu), by0 = wave_min_u32(in_arena ? oy : 0xFFFFFFFFu);
            const unsigned bx1 = wave_max_u32(in_arena ? ox : 0u), by1 = wave_max_u32(in_arena ? oy : 0u);
            if ((tid & 63) == 63) {
                atomicMin(&s_box[0], bx0); atomicMin(&s_box[1], by0);
                atomicMax(&s_box[2], bx1); atomicMax(&s_box[3], by1);
            }
        }
        __syncthreads();
        const float x0 = unord(s_box[0]), y0 = unord(s_box[1]);
        const float ex = unord(s_box[2]) - x0, ey = unord(s_box[3]) - y0;
        // cells a quarter of the neighbour range wide, or wider when 32 x 32 of them would not cover the arena
        const float cs = fmaxf(0.25f * p.neighbor_dist, fmaxf(ex, ey) * (1.0f / (GMAX - 0.5f)));
        const float ics = 1.0f / cs;
        gx0 = x0; gy0 = y0; gics = ics;
        Gx = min(GMAX, (int)(ex * ics) + 1); Gy = min(GMAX, (int)(ey * ics) + 1);
        auto cell_of = [&](V2 q2, int& ccx, int& ccy) {
            ccx = min(Gx - 1, max(0, (int)((q2.x - x0) * ics))); ccy = min(Gy - 1, max(0, (int)((q2.y - y0) * ics)));
        };
        int cx0, cy0;
        cell_of(pos0, cx0, cy0);
        int rank = 0;
        if (in_arena && h == 0) rank = atomicAdd(&s_ccnt[cy0 * Gx + cx0], 1);
        __syncthreads();
        if (tid < 64) {  // exclusive prefix sum over the cells: 16 cells per lane of the first wave
            constexpr int CPL = GMAX * GMAX / 64;
            int cnt[CPL];
            int sum = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) { cnt[k] = s_ccnt[CPL * tid + k]; sum += cnt[k]; }
            const int incl = wave_prefix_sum(sum);
            int b = incl - sum;
#pragma unroll
            for (int k = 0; k < CPL; ++k) { s_cstart[CPL * tid + k] = b; b += cnt[k]; }
            if (tid == 63) s_cstart[GMAX * GMAX] = incl;
        }
        __syncthreads();
        if (in_arena && h == 0) {
            const int dst = s_cstart[cy0 * Gx + cx0] + rank;
            s_sorted[dst] = (unsigned short)sl;
            s_sxy[dst] = make_float2(pos0.x, pos0.y);
        }
        __syncthreads();
        // ---- FROM HERE ON the pair of slot sl works for the agent at position sl of the SORTED list: a wave's 32 agents are
        // neighbours in space, so they walk the same cell rows (the candidate reads are LDS broadcasts, the loops of a wave
        // have one trip count), need blocks of cells of the same size, and need LP3 together or not at all ----
    }
    const int i = in_arena ? (int)s_sorted[sl] : sl;
    const int q = active ? a * N + i : 0;
    V2 pos = mk(s_px[i], s_py[i]), vel = mk(s_vx[i], s_vy[i]), pref = mk(0.0f, 0.0f);
    V2 pf32 = mk(1.0f, 0.0f);
    CA_PSTAMP(2);
    // ---- obstacle neighbours (App. A.2): brute force over the edge table, both lanes alike ----
    const double KEY_EMPTY = __longlong_as_double(0x7F800000FFFFFFFFll);  // (+inf, -1)
    const ObstDev* tab = p.obst + ((p.tab_off != nullptr && active) ? p.tab_off[a] : 0);  // this arena's edge table
    double okey[ST];
#pragma unroll
    for (int k = 0; k < ST; ++k) okey[k] = KEY_EMPTY;
    int oin = 0;
    {
        const float rangeSq = sqr(p.time_horizon_obst * p.max_speed + p.radius);
        auto visit = [&](const ObstDev& o1, int e, bool mine) __attribute__((always_inline)) {
            const V2 a1 = mk(o1.px, o1.py), a2 = mk(o1.qx, o1.qy);
            const float alol = leftOf(a1, a2, pos);
            const float dsl = div_ir(sqr(alol), absSq(a2 - a1));   // (an edge has a length; the quotient is only compared with the range)
            if (mine && dsl < rangeSq && alol < 0.0f) {
                const float dsq = distSqPointSegment(a1, a2, pos);
                if (dsq < rangeSq) {
                    ++oin;
                    sorted_insert_n<ST>(okey, make_key(dsq, e));
                }
            }
        };
        if (p.tab_off == nullptr) {  // one table for every arena: uniform loop, scalar loads of the edge records
            for (int e = 0; e < p.n_obst; ++e) visit(p.obst[e], e, active);
        } else {                     // a table per arena: ids are local to it
            const int ne = active ? p.tab_off[a + 1] - p.tab_off[a] : 0;
            for (int e = 0; __ballot(e < ne) != 0ull; ++e) {
                const bool mine = e < ne;
                visit(load_obst(tab, mine ? e : 0), e, mine);
            }
        }
    }
    const int ocnt = oin < S ? oin : S;

    // ---- agent neighbours (App. A.2): the candidates of every cell row dealt to the two lanes alternately ----
    double nkey[M];
#pragma unroll
    for (int k = 0; k < M; ++k) nkey[k] = KEY_EMPTY;
    {
        CA_PSTAMP(3);
        if (K > 0) {
            const float rangeSq0 = sqr(p.neighbor_dist);
            // HOW FAR TO LOOK.  The list this agent had after the last step names K other agents; wherever they are now, the
            // farthest of them bounds the distance of the K-th nearest agent now -- so the scan covers the cells that the disc
            // of that radius touches, not the whole neighbour range (a settled, contracted crowd: ~90 agents within the
            // neighbour range of 5, the tenth-nearest at 2.6 on average).  Any K distinct agents give a valid bound: nothing
            // depends on the list being fresh (after a reset it merely is a looser bound); a list the caller wrote through
            // ca_set is not trusted (p.nb_hint = 0), and a list shorter than K means the whole range.
            float b2 = rangeSq0;
            if (p.nb_hint && active && (int)(p.counts[q] & 0xFFu) == K) {
                float far2 = 0.0f;
                bool bad = false;
                int jn[KH];
                static_for<KH>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    jn[m] = (2 * m + h < K) ? ld_idx_t<CA_NBW16(BS)>(p.nb_idx, ((size_t)a * K + (2 * m + h)) * N + i) : -1;
                });
                static_for<KH>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    if (2 * m + h < K) {
                        const int j = jn[m];
                        const bool okj = j >= 0 && j < N && j != i;
                        const float d2 = absSq(pos - mk(s_px[okj ? j : i], s_py[okj ? j : i]));
                        far2 = d2 > far2 ? d2 : far2;
                        bad = bad || !okj;
                    }
                });
                const float of = quad_xor<0xB1>(far2);
                far2 = of > far2 ? of : far2;
                bad = bad || quad_xor<0xB1>(bad ? 1 : 0) != 0;
                if (!bad && far2 < rangeSq0) b2 = far2;
            }
            // the cells the closed disc of radius sqrt(b2) touches (+ 1e-4 relative and absolute: the cell index is a
            // monotone function of the coordinate, so a candidate within that distance lies in a cell between the cells of
            // the disc's extreme points)
            const float Bm = sqrtf(b2) * 1.0001f + 1e-4f;
            const int cxlo = max(0, (int)((pos.x - Bm - gx0) * gics)), cxhi = min(Gx - 1, (int)((pos.x + Bm - gx0) * gics));
            const int cylo = max(0, (int)((pos.y - Bm - gy0) * gics)), cyhi = min(Gy - 1, (int)((pos.y + Bm - gy0) * gics));
            for (int row = cylo; row <= (active ? cyhi : cylo - 1); ++row) {   // one run of the sorted list per cell row
                const int lo = s_cstart[row * Gx + cxlo], hi = s_cstart[row * Gx + cxhi + 1];
                int t = lo + h;
                int jn = 0;
                float2 on = make_float2(0.0f, 0.0f);
                if (t < hi) { jn = s_sorted[t]; on = s_sxy[t]; }
                while (t < hi) {
                    const int j = jn;
                    const V2 o = mk(on.x, on.y);
                    t += 2;
                    if (t < hi) { jn = s_sorted[t]; on = s_sxy[t]; }  // the next candidate is in flight during this one
                    const float dsq = absSq(pos - o);
                    const bool ok = j != i && dsq < rangeSq0;
                    sorted_insert_n<KMAX>(nkey, ok ? make_key(dsq, j) : KEY_EMPTY);
                }
            }
            merge_with_partner<M, 0xB1>(nkey);   // both lanes: the KMAX smallest keys of the two lists, ascending
        }
    }
    int ncnt = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) ncnt += (k < K && key_index(nkey[k]) >= 0) ? 1 : 0;

    CA_PSTAMP(4);
    // ---- the lists are state (the reference's reset() observes with the lists of the last doStep) ----
    if (active) {
        if (__builtin_expect(oin > S && h == 0, 0)) {
            atomicAdd(reinterpret_cast<int*>(&p.arena_stats[(size_t)a * ST_STRIDE + ST_OVERFLOW]), 1);
            note_overflow(p.cold, a, i, oin);
        }
        if (h == 0) p.counts[q] = (unsigned short)(ncnt | (ocnt << 8));
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if ((k & 1) == h && k < K) st_idx_t<CA_NBW16(BS)>(p.nb_idx, ((size_t)a * K + k) * N + i, key_index(nkey[k]));
#pragma unroll
        for (int k = 0; k < ST; ++k)
            if ((k & 1) == h && k < S) p.obst_idx[((size_t)a * S + k) * N + i] = (unsigned short)key_index(okey[k]);
    }

    // ---- obstacle ORCA lines (App. A.3), both lanes alike ----
    const float R = p.radius;
    float4 OB[ST];
    static_for<ST>([&](auto kc) __attribute__((always_inline)) { OB[decltype(kc)::value] = make_float4(0.0f, 0.0f, 1.0f, 0.0f); });
    int no = 0;
    if (__builtin_amdgcn_ballot_w64(ocnt > 0) != 0ull) {
        const float invTO = 1.0f / p.time_horizon_obst;
        static_for<ST>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            if (s < ocnt) {
                const int e = key_index(okey[s]);
                auto covered = [&](V2 c1, V2 c2) __attribute__((always_inline)) {
                    bool c = false;
                    static_for<ST>([&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        const Line Mj = unpack_line(OB[j]);
                        if (j < no && det(c1 - Mj.point, Mj.dir) - invTO * R >= -EPS &&
                            det(c2 - Mj.point, Mj.dir) - invTO * R >= -EPS)
                            c = true;
                    });
                    return c;
                };
                Line line;
                if (obst_orca_line(tab, e, pos, vel, R, invTO, covered, line)) {
                    const float4 pl = pack_line(line);
                    static_for<ST>([&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        if (j == no) OB[j] = pl;
                    });
                    ++no;
                }
            }
        });
    }
    float4 OBP[2];  // this lane's share of the obstacle lines in LP1: lines h and 2 + h
    OBP[0] = h ? OB[1] : OB[0];
    OBP[1] = h ? OB[3] : OB[2];
    CA_PSTAMP(5);
    // ---- agent ORCA lines (App. A.4): lane h builds the lines of neighbours h, h + 2, ... ----
    float4 LA[KH];
    static_for<KH>([&](auto kc) __attribute__((always_inline)) { LA[decltype(kc)::value] = make_float4(0.0f, 0.0f, 1.0f, 0.0f); });
    {
        const float invT = 1.0f / p.time_horizon;
        const float invDt = 1.0f / p.time_step;
        static_for<KH>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            int j = key_index(nkey[2 * m]);
            if constexpr (2 * m + 1 < M) { const int j1 = key_index(nkey[2 * m + 1]); j = h ? j1 : j; }
            if (2 * m + h < ncnt)
                LA[m] = pack_line(agent_orca_line(pos, vel, mk(s_px[j], s_py[j]), mk(s_vx[j], s_vy[j]), R, invT, invDt));
        });
    }
    CA_PSTAMP(6);
    // ---- 2-D linear program (App. A.5) ----
    const int nl = no + ncnt;
    V2 nv = mk(0.0f, 0.0f);
    int fail = nl;
    auto opt_fn = [&]() __attribute__((always_inline)) -> V2 {
        return mk(reinterpret_cast<const float*>(s_misc)[i * 4 + 2], reinterpret_cast<const float*>(s_misc)[i * 4 + 3]);
    };
    if (active) fail = lp2_pair<KMAX>(OB, OBP, LA, no, ncnt, h, p.max_speed, opt_fn, nv);
    CA_PSTAMP(7);
    // ---- LP3 for the agents whose LP2 was infeasible: the pair copies its lines into a slot of the wave's LDS pool and the
    // whole wave solves the slots, four lanes each (ca_lp.h lp3_coop).  (Dealing the arena's infeasible agents over the
    // pools of ALL its waves -- they cluster in the waves of the dense core -- was built and measured twice, before and
    // after the scan was shortened: the same kernel time, a round of lp3_coop is as long as its slowest slot whoever runs
    // it; profiles/r04_a_c5_phase_stamps_and_pair_kernel.txt.) ----
    {
        float4* pool = s_lines + (size_t)(tid >> 6) * (2 * ML) * POOL_SLOTS;
        float4* hdr = pool + (size_t)(2 * ML - 1) * POOL_SLOTS;
        bool need = active && fail < nl;
        const unsigned long long even = 0x5555555555555555ull;
        const unsigned long long below = (1ull << (tid & 62)) - 1ull;   // the lanes below this PAIR
        while (true) {
            const unsigned long long m = __ballot(need) & even;   // one bit per agent
            if (!m) break;
            const int rank = __popcll(m & below);                  // (the same value in both lanes of a pair)
            const bool mine = need && rank < POOL_SLOTS;
            if (mine) {
                float4* col = pool + rank;
                if (h == 0) {
                    static_for<ST>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k = decltype(kc)::value;
                        if (k < no) col[k * POOL_SLOTS] = OB[k];
                    });
                    hdr[rank] = make_float4(nv.x, nv.y, __int_as_float(nl | (no << 8) | (fail << 16)), 0.0f);
                }
                static_for<KH>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int mm = decltype(mc)::value;
                    if (2 * mm + h < ncnt) col[(no + 2 * mm + h) * POOL_SLOTS] = LA[mm];
                });
            }
            wave_lds_sync();
            const int waiting = __popcll(m);
            lp3_coop(pool, ML, waiting < POOL_SLOTS ? waiting : POOL_SLOTS, p.max_speed);  // the whole wave works
            wave_lds_sync();
            if (mine) {
                const float4 hv = hdr[rank];
                nv = mk(hv.x, hv.y);
                need = false;
            }
        }
    }
    if (active) {  // ---- integrate (App. A.1) ----
        vel = nv;
        pos = mk(s_px[i], s_py[i]) + vel * p.time_step;  // own pre-step position: still in the staged arena
    }
    CA_PSTAMP(8);
    // ---- epilogue (ca_step.h, same order of operations; lane 0 of a pair writes) ----
    typedef const __attribute__((address_space(4))) StepCold ColdP;
    const ColdP& c = *(ColdP*)p.cold;
    pf32 = mk(reinterpret_cast<float*>(s_misc)[i * 4 + 0], reinterpret_cast<float*>(s_misc)[i * 4 + 1]);
    pref = mk(reinterpret_cast<float*>(s_misc)[i * 4 + 2], reinterpret_cast<float*>(s_misc)[i * 4 + 3]);
    {   // how far does any agent of the arena move in this step?
        const unsigned sp = wave_max_u32(active ? __float_as_uint(absSq(vel)) : 0u);
        if ((tid & 63) == 63) atomicMax(&s_vmax2, sp);
    }
    __syncthreads();  // every lane is done with the pre-step arena image
    CA_PSTAMP(9);
    if (h == 0) { s_px[i] = pos.x; s_py[i] = pos.y; }
    int* red = s_red;  // [0] not-done agents, [1] pairs, [2] wall hits, [3] goals (cleared at the head of the kernel)
    __syncthreads();

    if (p.flags & 2u) {  // CA_F_STATS: see ca_step.h for why K distances replace the scan of the arena
        int pairs = 0;
        const float crSq = sqr(R + R);
        // (2 m of ca_step.h's argument, m = the arena's largest speed of this step x dt, measured: a bound from max_speed is not safe)
        const float m2 = 2.0002f * __builtin_sqrtf(__uint_as_float(s_vmax2)) * p.time_step;
        bool scan_all = active && !(p.neighbor_dist >= R + R + m2);
        if (active && !scan_all) {
            float far2 = 0.0f;
            int jn[KH];  // this lane's list entries, read back (the lane wrote them itself): not kept in registers across the solve
            static_for<KH>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                jn[m] = (2 * m + h < ncnt) ? ld_idx_t<CA_NBW16(BS)>(p.nb_idx, ((size_t)a * K + (2 * m + h)) * N + i) : 0;
            });
            static_for<KH>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                const int j = jn[m];
                if (2 * m + h < ncnt) {
                    const float d2 = absSq(pos - mk(s_px[j], s_py[j]));
                    far2 = d2 > far2 ? d2 : far2;
                    if (j > i && d2 < crSq) ++pairs;
                }
            });
            const float of = quad_xor<0xB1>(far2);
            far2 = of > far2 ? of : far2;
            scan_all = (ncnt == K) && !(far2 > sqr(R + R + 2.0f * m2));
        }
        if (__ballot(scan_all) != 0ull && scan_all) {
            pairs = 0;
            for (int j = i + 1 + h; j < N; j += 2)
                if (absSq(pos - mk(s_px[j], s_py[j])) < crSq) ++pairs;
        }
        pairs = pair_sum(pairs);
        if (active) {
            bool wall = false;
            if (p.tab_off == nullptr) {
                for (int e = 0; e < p.n_obst; ++e) {
                    const ObstDev o1 = p.obst[e];
                    if (distSqPointSegment(mk(o1.px, o1.py), mk(o1.qx, o1.qy), pos) < sqr(R)) wall = true;
                }
            } else {
                const int ne = p.tab_off[a + 1] - p.tab_off[a];
                for (int e = 0; e < ne; ++e) {
                    const ObstDev o1 = load_obst(tab, e);
                    if (distSqPointSegment(mk(o1.px, o1.py), mk(o1.qx, o1.qy), pos) < sqr(R)) wall = true;
                }
            }
            if (h == 0) {
                if (pairs) atomicAdd(&red[1], pairs);
                if (wall) atomicAdd(&red[2], 1);
            }
        }
    }

    CA_PSTAMP(10);
    // ---- reward (env.py:389-400) or preferred velocity towards the goal (env.py:449) ----
    float rew = 0.0f;
    double gx = 0.0, gy = 0.0;
    if (active) {
        gx = c.goal_x[q]; gy = c.goal_y[q];
        if (p.actions) {
            const float scale = (float)c.reward_scale;
            const float r_goal = vel.x * pf32.x + vel.y * pf32.y;
            const float r_polite = vel.x * pref.x + vel.y * pref.y;  // pref still is the action-rotated direction (env.py:381)
            rew = scale * r_goal + (1.0f - scale) * r_polite;
            if (h == 0) c.reward[q] = rew;
        } else {
            double dx, dy;
            pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
            pref = mk((float)dx, (float)dy);
        }
    }
    // ---- step counter and done test (env.py:352-365, 404-410; ALAN:118-121, 547-566) ----
    const bool nodone = (p.flags & 8u) != 0;  // CA_F_NODONE
    bool goal_changed = false;
    int done = active ? c.agent_done[q] : 1;
    int steps = active ? c.step_count[a] : 0;
    if (!p.actions && !nodone) ++steps;
    if (active && !nodone) {
        bool hit = false;
        if (c.done_mode == 0) {
            hit = (done == 0) && (pos.x < c.done_x_thresh);
        } else {
            const double dx = (double)pos.x - gx, dy = (double)pos.y - gy;
            const double lim = 2.0 * (double)p.radius;
            hit = (dx * dx + dy * dy) < lim * lim;
            if (c.done_mode == 1) hit = hit && (done == 0);
        }
        if (hit) {
            if (c.done_mode == 2) {
                const int rc = c.regoal_count[q];
                double u0, u1;
                rng2(c.seed, c.arena_offset + a, i, RNG_REGOAL, (uint32_t)rc, &u0, &u1);
                gx = uniform64((double)c.goal_x0, (double)c.goal_x1, u0);
                gy = uniform64((double)c.goal_y0, (double)c.goal_y1, u1);
            } else {
                done = 1;
                gx = c.goal2_x[q]; gy = c.goal2_y[q];
            }
            goal_changed = true;
            if (h == 0) atomicAdd(&red[3], 1);
        }
    }
    if (p.actions) ++steps;
    if (active && done == 0 && h == 0) atomicAdd(&red[0], 1);
    __syncthreads();   // (both lanes of a pair have read regoal_count / agent_done / the goal before lane 0 rewrites them below)
    if (goal_changed && h == 0) {
        if (c.done_mode == 2) {
            c.regoal_count[q] = c.regoal_count[q] + 1;
        } else {
            c.arrive_step[q] = steps;
            c.agent_done[q] = 1;
        }
        c.goal_x[q] = gx; c.goal_y[q] = gy;
    }

    bool all_done = false;
    if (active) {
        all_done = !nodone && (red[0] == 0);
        if (c.max_step > 0 && steps >= c.max_step) all_done = true;
    }
    const bool do_reset = all_done && (p.flags & 4u);  // CA_F_AUTORESET
    int epi = 0;
    if (do_reset) {  // env.py:461-488 for this arena
        epi = c.episode[a];
        double u0, u1;
        rng2(c.seed, c.arena_offset + a, i, RNG_RESET, (uint32_t)epi, &u0, &u1);
        pos = mk((float)uniform64((double)c.spawn_x0, (double)c.spawn_x1, u0),
                 (float)uniform64((double)c.spawn_y0, (double)c.spawn_y1, u1));
        done = 0;
        double dx, dy;
        pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
        pref = mk((float)dx, (float)dy);
    }
    // sum of rewards: a fixed-shape tree over the wave's agents (lane 0 of every pair carries the value)
    if (p.actions && (p.flags & 2u)) {
        double r = (active && h == 0) ? (double)rew : 0.0;
        for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
        if ((tid & 63) == 0 && a < p.a1 && !frozen && (tid >> 1) < N)
            atomicAdd(reinterpret_cast<double*>(&c.arena_stats[(size_t)a * ST_STRIDE + ST_SUMREW]), r);
    }
    // orientation of the observation frame (env.py:236): direction to the goal from the final state
    float ox = pref.x, oy = pref.y;
    if (active && !do_reset && (p.actions != nullptr || goal_changed)) {
        double dx, dy;
        pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
        ox = (float)dx; oy = (float)dy;
    }
    __syncthreads();  // all lanes have read red[] and episode[]
    if (active && h == 0) {
        if (do_reset) c.agent_done[q] = 0;
        c.orient_x[q] = ox; c.orient_y[q] = oy;
        c.pos_x[q] = pos.x; c.pos_y[q] = pos.y;
        c.vel_x[q] = vel.x; c.vel_y[q] = vel.y;
        c.pref_x[q] = pref.x; c.pref_y[q] = pref.y;
        if (i == 0) {
            unsigned long long* st = c.arena_stats + (size_t)a * ST_STRIDE;
            if (red[1]) st[ST_COLL] += (unsigned)red[1];
            if (red[2]) st[ST_OBST_COLL] += (unsigned)red[2];
            if (red[3]) st[ST_GOALS] += (unsigned)red[3];
            if (all_done) {
                st[ST_EPISODES] += 1;
                st[ST_LASTEP] = ((unsigned long long)(unsigned)steps << 32) | (unsigned)(N - red[0]);
            }
            c.arena_done[a] = all_done ? 1 : 0;
            c.step_count[a] = do_reset ? 0 : steps;
            if (do_reset) c.episode[a] = epi + 1;
            atomicAdd(&c.arena_steps[a], 1ull);
        }
    }
    CA_PSTAMP(11);
}

}  // namespace ca
