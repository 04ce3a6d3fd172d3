// ca_nbr.h -- neighbour search (App. A.2): sorted key lists, brute-force and uniform-grid scans
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
#pragma once
#include "ca_common.h"

namespace ca {

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
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int MAXN>
__device__ __forceinline__ void sorted_insert(double (&key)[MAXN], double x) {
#pragma unroll
    for (int k = MAXN - 1; k >= 1; --k) key[k] = key_max(key[k - 1], key_min(key[k], x));
    key[0] = key_min(key[0], x);
}

// ============================================================================================
// Neighbour search for every agent (App. A.2): the obstacle edges within range and
// the K nearest agents, written as the lists [A,S,N] / [A,K,N] that the solve kernel and the
// observation read.  A kernel of its own because it needs almost no LDS (the arena's positions,
// 8 B per lane): it runs at full occupancy and is issue-bound, whereas the solve kernel is tied to
// its 16 B x (K+S) line table per lane.
// ============================================================================================
// SM: capacity of the register list of obstacle neighbours (S <= SM): 4 for the register-line solve, SMAX for the LDS table
// HELP = 2: the workgroup is launched with 2 BS lanes; lanes BS .. 2 BS - 1 are HELPERS of agents 0 .. BS - 1 for the
// uniform-grid scan only (large arenas: the scan is the longest dependent chain of the step and a 512-agent arena is
// just 8 waves on its CU): main lane and helper take alternate candidates of every cell row, each keeping its K nearest,
// the helper hands its list over through LDS and ends; the main lane inserts it into its own (keys are totally ordered:
// the K smallest of the union are the serial scan's list).  Returns true for a helper lane (the caller returns too).
template <int KMAX, int BS, int SM, int HELP = 1>
__device__ __forceinline__ bool nbr_body(const StepArgs& p) {
#ifndef CA_NBR_NO_VGPR_PAD
    // Claim 128 VGPRs (the kernel needs 56): at most 4 waves then fit on a SIMD, so a launch that brings one
    // wave per SIMD slot (4096 x 64 lanes on 256 CUs) is spread evenly.  Without it the dispatcher puts
    // anything from 1 to 7 of these light waves on a SIMD and the kernel waits for the fullest one.
    asm volatile("" ::: "v127");
#endif
    __shared__ float s_px[BS];
    __shared__ float s_py[BS];
    const int tid0 = threadIdx.x;
    const bool helper = HELP > 1 && tid0 >= BS;   // (whole waves: BS is a multiple of 64)
    const int tid = helper ? tid0 - BS : tid0;    // the agent slot this lane works for
    const int P = p.P;
    int la, i;
    lane_slot(p, tid, la, i);
    const int apb = p.apb;
    const int a = p.a0 + work_block(p) * apb + la;
    const bool active = (a < p.a1) && (i < p.N) && (la < apb) && !arena_frozen(p, a);
    const int N = p.N, K = p.K, S = p.S;
    const int q = active ? a * N + i : 0;
    const int lbase = tid - i;
    CA_STAMP(12);
    V2 pos = mk(0.0f, 0.0f);
    if (active && !helper) pos = mk(p.pos_x[q], p.pos_y[q]);
    if (!helper) { s_px[tid] = pos.x; s_py[tid] = pos.y; }
    __syncthreads();
    if (helper) pos = mk(s_px[tid], s_py[tid]);

    const float INF = __int_as_float(0x7f800000);
    // ---- obstacle neighbours (App. A.2): brute force over the edge table ----
    const int sofs = SM - S;  // the S-entry list is right-aligned in the register array
    const double KEY_EMPTY = __longlong_as_double(0x7F800000FFFFFFFFll);  // (+inf, -1)
    const double KEY_DUMMY = __longlong_as_double((long long)0xFFF0000000000000ull);  // -inf: never moves
    double okey[SM];
#pragma unroll
    for (int k = 0; k < SM; ++k) okey[k] = (k < sofs) ? KEY_DUMMY : KEY_EMPTY;
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
                    sorted_insert<SM>(okey, make_key(dsq, e));
                }
            }
        };
        if (helper) {                // (helpers work in the agent scan only)
        } else if (p.tab_off == nullptr) {  // one table for every arena: uniform loop, scalar loads of the edge records
            for (int e = 0; e < p.n_obst; ++e) visit(p.obst[e], e, active);
        } else {                     // a table per arena (several arenas may share this wave): ids are local to it
            const int t0 = active ? p.tab_off[a] : 0, ne = active ? p.tab_off[a + 1] - t0 : 0;
            for (int e = 0; __ballot(e < ne) != 0ull; ++e) {
                const bool mine = e < ne;
                visit(load_obst(p.obst, mine ? t0 + e : 0), e, mine);
            }
        }
    }
    const int ocnt = oin < S ? oin : S;
    if constexpr (SM > 4) {   // a list of 16 keys is 32 registers: it goes to memory now, not after the agent scan
        if (active && !helper) {
#pragma unroll
            for (int k = 0; k < SM; ++k)
                if (k >= sofs) p.obst_idx[((size_t)a * S + (k - sofs)) * N + i] = (unsigned short)key_index(okey[k]);
        }
    }
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
        // the block of cells around it -- one contiguous run of the sorted list per cell row -- instead of the whole arena.
        // The cells are visited in no particular index order, so a candidate enters on `distance <= current
        // K-th distance` and the 64-bit (distance, index) keys settle ties; the list is the same K smallest
        // keys within neighbor_dist that the index-order scan keeps.
        if (P == BS && N >= 192 && K > 0) {
            constexpr int GMAX = BS >= 1024 ? 16 : 32;  // at most GMAX x GMAX cells (the 1024-lane shape has no LDS to spare)
            __shared__ unsigned s_box[4];          // ordered-uint images of min x, min y, max x, max y
            __shared__ int s_ccnt[GMAX * GMAX];    // agents per cell
            __shared__ int s_cstart[GMAX * GMAX + 1];  // first position of a cell in s_sorted
            __shared__ unsigned short s_sorted[BS];
            auto ord = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
            auto unord = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); };
            if (tid0 < 2) s_box[tid0] = 0xFFFFFFFFu;
            if (tid0 >= 2 && tid0 < 4) s_box[tid0] = 0u;
            for (int cidx = tid0; cidx < GMAX * GMAX; cidx += BS * HELP) s_ccnt[cidx] = 0;
            __syncthreads();
            const bool in_arena = (a < p.a1) && (i < N) && !helper;  // frozen arenas skip the scan but keep the barriers
            {  // the arena's bounding box: a reduction per wave, then one atomic per wave and corner
                const unsigned ox = ord(pos.x), oy = ord(pos.y);
                const unsigned bx0 = wave_min_u32(in_arena ? ox : 0xFFFFFFFFu), by0 = wave_min_u32(in_arena ? oy : 0xFFFFFFFFu);
                const unsigned bx1 = wave_max_u32(in_arena ? ox : 0u), by1 = wave_max_u32(in_arena ? oy : 0u);
                if ((tid0 & 63) == 63) {
                    atomicMin(&s_box[0], bx0); atomicMin(&s_box[1], by0);
                    atomicMax(&s_box[2], bx1); atomicMax(&s_box[3], by1);
                }
            }
            __syncthreads();
            // Cells half a neighbour range wide (the 5 x 5 block around an agent's cell covers its range with 156 / 225 of
            // the area of 3 x 3 cells a full range wide: a third fewer candidates), or wider when the arena is so large that
            // 32 x 32 of them would not cover it; RC = cells to either side that can hold a neighbour (1 or 2).
            const float x0 = unord(s_box[0]), y0 = unord(s_box[1]);
            const float ex = unord(s_box[2]) - x0, ey = unord(s_box[3]) - y0;
            const float cs = fmaxf((GMAX >= 32 ? 0.5f : 1.0f) * p.neighbor_dist, fmaxf(ex, ey) * (1.0f / (GMAX - 0.5f)));
            const int RC = (cs >= p.neighbor_dist) ? 1 : 2;
            const float ics = 1.0f / cs;
            const int Gx = min(GMAX, (int)(ex * ics) + 1), Gy = min(GMAX, (int)(ey * ics) + 1);
            const int cx = min(Gx - 1, max(0, (int)((pos.x - x0) * ics))), cy = min(Gy - 1, max(0, (int)((pos.y - y0) * ics)));
            int rank = 0;
            if (in_arena) rank = atomicAdd(&s_ccnt[cy * Gx + cx], 1);
            __syncthreads();
            if (tid0 < 64) {  // exclusive prefix sum over the cells: GMAX^2 / 64 cells per lane of the first wave
                constexpr int CPL = GMAX * GMAX / 64;
                int cnt[CPL];
                int sum = 0;
#pragma unroll
                for (int k = 0; k < CPL; ++k) { cnt[k] = s_ccnt[CPL * tid0 + k]; sum += cnt[k]; }
                int incl = sum;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off);
                    if (tid0 >= off) incl += t;
                }
                int b = incl - sum;
#pragma unroll
                for (int k = 0; k < CPL; ++k) { s_cstart[CPL * tid0 + k] = b; b += cnt[k]; }
                if (tid0 == 63) s_cstart[GMAX * GMAX] = incl;
            }
            __syncthreads();
            // (up to 512 lanes the positions are sorted along with the indices: a candidate then is one LDS address, known
            // an iteration ahead, instead of an index and a dependent gather; the 1024-lane shape has no LDS for that)
            constexpr bool SXY = BS <= 512;
            __shared__ float2 s_sxy[SXY ? BS : 1];
            if (in_arena) {
                const int dst = s_cstart[cy * Gx + cx] + rank;
                s_sorted[dst] = (unsigned short)i;
                if constexpr (SXY) s_sxy[dst] = make_float2(pos.x, pos.y);
            }
            __syncthreads();
            const float rangeSq0 = sqr(p.neighbor_dist);
            float rangeK = rangeSq0;  // distance of the current K-th entry once the list is full
            auto row_range = [&](int ry, int& lo, int& hi) {
                const int row = cy + ry;
                lo = 0; hi = 0;
                if (active && row >= 0 && row < Gy) {
                    lo = s_cstart[row * Gx + max(cx - RC, 0)];
                    hi = s_cstart[row * Gx + min(cx + RC, Gx - 1) + 1];
                }
            };
            auto visit = [&](int j, const V2& o) {
                const float dsq = absSq(pos - o);
                if (j != i && dsq < rangeSq0 && dsq <= rangeK) {
                    sorted_insert<KMAX>(nkey, make_key(dsq, j));
                    if (ncnt < K) ++ncnt;
                    if (ncnt == K) rangeK = key_dist(nkey[KMAX - 1]);
                }
            };
            int lo_n, hi_n;
            row_range(-RC, lo_n, hi_n);
            for (int ry = -RC; ry <= RC; ++ry) {
                const int lo = lo_n, hi = hi_n;
                if (ry < RC) row_range(ry + 1, lo_n, hi_n);  // the next row's bounds are in flight during this row
                int t = lo + (helper ? 1 : 0);
                if constexpr (SXY) {
                    int jn = 0;
                    float2 on = make_float2(0.0f, 0.0f);
                    if (t < hi) { jn = s_sorted[t]; on = s_sxy[t]; }
                    while (t < hi) {
                        const int j = jn;
                        const V2 o = mk(on.x, on.y);
                        t += HELP;
                        if (t < hi) { jn = s_sorted[t]; on = s_sxy[t]; }  // the next candidate is in flight during this one
                        visit(j, o);
                    }
                } else {
                    for (; t < hi; t += HELP) {
                        const int j = s_sorted[t];
                        visit(j, mk(s_px[j], s_py[j]));
                    }
                }
            }
            if constexpr (HELP > 1) {  // the helper's list -> LDS -> the main lane's list
                __shared__ double s_hkey[KMAX][BS];
                if (helper) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) s_hkey[k][tid] = nkey[k];
                }
                __syncthreads();
                if (helper) return true;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k >= kofs) sorted_insert<KMAX>(nkey, s_hkey[k][tid]);  // (an empty slot is the largest key: no effect)
                ncnt = 0;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) ncnt += (k >= kofs && key_index(nkey[k]) >= 0) ? 1 : 0;
            }
            scanned = true;
        }
    }
#ifndef CA_CK_MAXN
#define CA_CK_MAXN 64    // the largest arena that first tries the composite keys (see below: larger arenas measured slower)
#endif
    if (K > 0 && !scanned && N <= CA_CK_MAXN) {
        // Arenas of at most 64 agents: 32-bit composite keys = (distance image << logP) | candidate index and
        // one v_med3_u32 per list slot and candidate -- new[k] = med3(old[k-1], old[k], x) IS the sorted insert,
        // at half the instructions of the 64-bit network (in so small an arena some lane accepts nearly every
        // candidate, so the shrinking range of the contract never lets a wave skip the network anyway).
        // The image is the fp32 BIT PATTERN of the squared distance, counted down from that of neighbor_dist^2
        // (round 5): 32 - logP bits hold every float of the top 2^(9 - logP) binades below the range -- for 64 agents and a range
        // of 5 every d2 in [0.125, 25) -- at full resolution, so there the composite order IS the contract's exact (distance, index)
        // order (equal images = equal distances = index order), and everything nearer than that is clamped to image 0.  Only a
        // lane with two or more such candidates (agents within 0.35 of each other at radius 0.5: deep overlaps) cannot order
        // them; a wave in which some lane has them falls through to the exact 64-bit scan below.  (Rounds 2-4 used a fixed-point
        // image floor(d2 * 2^26 / 25): a settled crowd packs at exactly the contact distance, the neighbours of an agent in a
        // dense core differ by a few ulps of d2 = 1 -- less than that image resolves -- and every fifth wave scanned twice:
        // profiles/r05_c_exact_composite_keys.txt.)
        const float rangeSq0 = sqr(p.neighbor_dist);
        const unsigned lowmask = (unsigned)(P - 1);
        const unsigned top = __float_as_uint(rangeSq0);                  // a passing candidate's bits lie below it
        const unsigned span = 0xFFFFFFFFu >> p.logP;                     // images 0 .. span - 2 (the composite never is ~0)
        // bit patterns up to u0 share image 0 (arenas of at most four lanes with a range below 1: every pattern fits, u0 = 0)
        const unsigned u0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(top >= span ? top - span + 1u : 0u));
        unsigned ck[KMAX + 1];
#pragma unroll
        for (int k = 0; k <= KMAX; ++k) ck[k] = (k < kofs) ? 0u : 0xFFFFFFFFu;
        auto visit = [&](int j, V2 o) __attribute__((always_inline)) {
            const float dsq = absSq(pos - o);
            const bool pass = active && j != i && dsq < rangeSq0;
            const unsigned u = __float_as_uint(dsq);
            unsigned key = (((u > u0 ? u : u0) - u0) << p.logP) | (unsigned)j;
            asm volatile("" : "+v"(key));   // (computed for every lane: left to itself the compiler branches around these three
            const unsigned c = pass ? key : 0xFFFFFFFFu;   // instructions, twice per candidate; now one v_cndmask)
#pragma unroll
            for (int k = KMAX; k >= 1; --k) ck[k] = umed3(ck[k - 1], ck[k], c);
            ck[0] = ck[0] < c ? ck[0] : c;
        };
        // four candidates per trip, their positions read at the head of the trip (immediate LDS offsets, no register rotation
        // between trips: the rolled loop with a one-ahead prefetch spent 5 of its 28 vector instructions per candidate on moves
        // and address increments)
        int j = 0;
        for (; j + 4 <= N; j += 4) {
            const V2 o0 = mk(s_px[lbase + j], s_py[lbase + j]), o1 = mk(s_px[lbase + j + 1], s_py[lbase + j + 1]);
            const V2 o2 = mk(s_px[lbase + j + 2], s_py[lbase + j + 2]), o3 = mk(s_px[lbase + j + 3], s_py[lbase + j + 3]);
            visit(j, o0); visit(j + 1, o1); visit(j + 2, o2); visit(j + 3, o3);
        }
        for (; j < N; ++j) visit(j, mk(s_px[lbase + j], s_py[lbase + j]));
        int cnt = 0;
        bool clamped_pair = false;   // (the list is ascending: two clamped candidates, if there are any, are its first two entries)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k >= kofs && ck[k] != 0xFFFFFFFFu) ++cnt;
            if (k == kofs) clamped_pair = ck[k + 1] <= lowmask;
        }
        if (__builtin_amdgcn_ballot_w64(clamped_pair) == 0ull) {  // wave-uniform
            ncnt = cnt;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)  // hand the indices over in the key array the store below reads
                nkey[k] = (k < kofs) ? KEY_DUMMY : make_key(0.0f, (ck[k] == 0xFFFFFFFFu) ? -1 : (int)(ck[k] & lowmask));
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
        if (__builtin_expect(oin > S, 0)) {
            atomicAdd(reinterpret_cast<int*>(&p.arena_stats[(size_t)a * ST_STRIDE + ST_OVERFLOW]), 1);
            note_overflow(p.cold, a, i, oin);
        }
        p.counts[q] = (unsigned short)(ncnt | (ocnt << 8));
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k >= kofs) st_idx_t<CA_NBW16(BS)>(p.nb_idx, ((size_t)a * K + (k - kofs)) * N + i, key_index(nkey[k]));  // ids of <= 256 agents fit a byte
        if constexpr (SM <= 4) {
#pragma unroll
            for (int k = 0; k < SM; ++k)
                if (k >= sofs) p.obst_idx[((size_t)a * S + (k - sofs)) * N + i] = (unsigned short)key_index(okey[k]);
        }
    }
    CA_STAMP(15);
    return false;
}

template <int KMAX, int BS, int SM>
__global__ __launch_bounds__(BS) void nbr_kernel(const StepArgs p) {
    nbr_body<KMAX, BS, SM>(p);
}

}  // namespace ca
