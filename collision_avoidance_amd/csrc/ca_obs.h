// ca_obs.h -- the laser observation kernel
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
#pragma once
#include "ca_common.h"

namespace ca {

// ============================================================================================
// Laser observation (env.py:231-318, utils.py:5-113).
//
// A workgroup (256 lanes) owns 16 agents of ONE arena; 16 lanes per agent.  The arena's
// positions/velocities are staged in LDS once (the neighbour gathers then never leave the CU).
// Phase A -- lane per (source, ray) pair (8 octagon chords per ORCA agent neighbour, one segment per ORCA
//   obstacle neighbour): the segment is rotated into the goal-aligned frame and tested only against the
//   rays that can possibly reach it: the rays inside its angular span as seen from the origin (a
//   conservative superset, see the pre-pass).  The (neighbour, ray) pairs of the workgroup's 16 agents
//   form ONE list walked by all its lanes, so a wave takes a second trip only for what does not fit the
//   whole workgroup.  The ray/segment test itself is the reference's arithmetic, so culling never
//   changes a result.  A hit is merged into the ray's slot with one LDS ds_min_u64 on the key
//   (distance bits << 32 | segment index): the minimum distance wins and equal distances resolve
//   to the first segment, exactly like a serial first-minimum scan.
// Phase B -- lane per RAY: re-derives the winning segment's hit point and velocity and writes its
//   4 floats; the 16 lanes of an agent write its 256-B row, a wave stores 1 KiB contiguously.
// ============================================================================================
struct ObsArgs {
    const float *pos_x, *pos_y, *vel_x, *vel_y, *orient_x, *orient_y;
    const unsigned short* counts;       // packed like StepArgs
    const void* nb_idx;                 // u8, or u16 when the arena has more than 256 agents (template parameter NW16)
    const unsigned short* obst_idx;
    const ObstDev* obst;
    const int* tab_off;                 // null or [A + 1]: per-arena edge tables (see StepArgs)
    float* obs;
    int A, N, K, S, bpa;  // bpa = workgroups per arena = ceil(N / 16)
    int nstage_max;       // entries of the staged position / velocity arrays (sizes the LDS carve-up)
    int paircap;          // entries of an agent's (source, ray) pair list: 16 rays x (K + S) sources
    int a0;               // first arena of this launch
    int xcd;              // 1: workgroup b serves an arena with (arena - a0) % 8 == b % 8, i.e. on the XCD whose solve wave wrote it
    int dense;            // 1 (arenas of fewer than 16 agents): a workgroup's 16 agent groups are 16 CONSECUTIVE agents of the batch,
                          // whatever arenas they belong to -- the reference env's own 10-agent arenas fill 16 of 16 groups instead of 10
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

// LDS (bytes): arena px,py,vx,vy [N] | keys [16][16] u64 | neighbour positions relative to the agent [16][16] float2 | agent frames [16] float4 | nb idx [16][16] | obstacle idx [16][16]
//              | ray and octagon tables [64] | pair counts [2][16] | (source, ray) pairs [16 x 16 (K + S)] u16: the
//              workgroup's (agent, neighbour, ray) list from the front, every agent's obstacle pairs in a block of its
//              own from the back
__host__ __device__ inline size_t obs_lds_bytes(int nstage_max, int obs_bs, int paircap) {
    const size_t apb = obs_bs / 16;
    return (size_t)nstage_max * 16 + 2 * apb * 16 * 8 + apb * 16 + apb * 16 * 4 + apb * 16 * 4 + 64 * 4 + 2 * apb * 4 + apb * (size_t)paircap * 2;
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
template <int OBS_BS, bool NW16, bool DENSE = false>
__global__ __launch_bounds__(OBS_BS) void obs_kernel(const ObsArgs p) {
    constexpr int OBS_APB = OBS_BS / 16;  // agents per workgroup
    extern __shared__ float4 smem4[];
    const int tid = threadIdx.x;
    const int g = tid >> 4, r = tid & 15;
    const int N = p.N, K = p.K, S = p.S;
    const int PAIRCAP = p.paircap;
    // Workgroups go round the 8 XCDs by index, and an arena's state was written by solve workgroup `arena` (one arena per
    // solve workgroup from 16 agents up): its observation workgroups take indices of the same residue, so that what they
    // read sits in that XCD's L2 (C3: obs_kernel 62.2 -> 61.5 us).
    int bid = blockIdx.x;
    if (p.xcd) { const int x = bid & 7, idx = bid >> 3, q8 = idx / p.bpa; bid = (x + 8 * q8) * p.bpa + (idx - q8 * p.bpa); }
    int a, i, a_lo, nstage;   // this group's arena and agent; first arena and number of agents the workgroup stages
    bool active;
    if constexpr (DENSE) {   // groups = consecutive agents of the batch
        const int first = bid * OBS_APB, gi = first + g;
        active = gi < p.A * N;
        const int ga = active ? gi : p.A * N - 1;
        a = p.a0 + ga / N; i = ga - (ga / N) * N;
        a_lo = p.a0 + first / N;
        const int last = min(first + OBS_APB - 1, p.A * N - 1);
        nstage = (last / N - first / N + 1) * N;
    } else {
        const int ab = bid / p.bpa;
        a = p.a0 + ab;
        i = (bid - ab * p.bpa) * OBS_APB + g;
        active = i < N;
        a_lo = a; nstage = N;
    }
    const int abase = DENSE ? (a - a_lo) * N : 0;   // LDS index of agent 0 of this group's arena
    const size_t q = (size_t)a * N + (active ? i : 0);
    // One arena per workgroup (not DENSE): the arena's part of every address is the same for all lanes -- a scalar base -- and a
    // lane adds a 32-bit offset inside the arena, so the loads and the row store below take the "scalar base + vector offset" form
    // and no lane does 64-bit address arithmetic (it was 35 of the wave's ~575 vector instructions).
    const size_t aoff = (size_t)a * N;
    const unsigned ii = active ? (unsigned)i : 0u;
    const ObstDev* tab = p.obst + (p.tab_off ? p.tab_off[a] : 0);  // this arena's edge table

    const int NST = p.nstage_max;   // agents staged at most (N; dense: up to 16 + 2 N)
    float* s_px = reinterpret_cast<float*>(smem4);
    float* s_py = s_px + NST;
    float* s_vx = s_py + NST;
    float* s_vy = s_vx + NST;
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_vy + NST);  // NST*16 B: 8-aligned
    float2* s_rel = reinterpret_cast<float2*>(s_key + OBS_APB * 16);             // neighbour slot k of agent g: p_nb - p_g
    float4* s_frame = reinterpret_cast<float4*>(s_rel + OBS_APB * 16);           // (cos, sin, pos x, pos y) per agent
    int* s_nb = reinterpret_cast<int*>(s_frame + OBS_APB);
    int* s_ob = s_nb + OBS_APB * 16;
    float* s_rays = reinterpret_cast<float*>(s_ob + OBS_APB * 16);  // [32] rays then [32] octagon
    float* s_oct = s_rays + 32;
    int* s_cnt = reinterpret_cast<int*>(s_oct + 32);                       // [0]: neighbour pairs of the workgroup
    int* s_cnt2 = s_cnt + OBS_APB;                                         // [16] obstacle pairs per agent
    unsigned short* s_pair = reinterpret_cast<unsigned short*>(s_cnt2 + OBS_APB);  // [16 * paircap]
    const int SPAIRS = 16 * S;  // an agent's block of obstacle pairs
    CA_OSTAMP(0);
    if (tid < 32) { s_rays[tid] = p.rays[tid]; s_oct[tid] = p.oct[tid]; }

    // Arenas of more than 256 agents (16-bit neighbour ids) are not staged: each of the arena's N / 16 workgroups would
    // copy all N agents (C5: 32 workgroups x 8 KB, a quarter of a wave's cycles) to look up 16 x K of them -- the neighbours'
    // positions and velocities are gathered from the arena's global arrays (L2-resident: the solve has just written them).
    constexpr bool GATHER = NW16;
    const float* gpx = p.pos_x + (size_t)a * N;
    const float* gpy = p.pos_y + (size_t)a * N;
    const float* gvx = p.vel_x + (size_t)a * N;
    const float* gvy = p.vel_y + (size_t)a * N;
    if constexpr (!GATHER) {
        if constexpr (DENSE) {
            for (int t = tid; t < nstage; t += OBS_BS) {   // (the staged arenas are contiguous in memory)
                const size_t qa = (size_t)a_lo * N + t;
                s_px[t] = p.pos_x[qa]; s_py[t] = p.pos_y[qa]; s_vx[t] = p.vel_x[qa]; s_vy[t] = p.vel_y[qa];
            }
        } else {
            for (unsigned t = tid; t < (unsigned)nstage; t += OBS_BS) {
                s_px[t] = (p.pos_x + aoff)[t]; s_py[t] = (p.pos_y + aoff)[t]; s_vx[t] = (p.vel_x + aoff)[t]; s_vy[t] = (p.vel_y + aoff)[t];
            }
        }
    }
    int nn = 0, ns = 0;
    float c = 1.0f, s = 0.0f;
    if (active) {
        if constexpr (DENSE) {
            const int cnts = p.counts[q];
            nn = cnts & 0xFF; ns = cnts >> 8;
            c = p.orient_x[q]; s = -p.orient_y[q];  // utils.py:48-51: cos/sin of -atan2(orientation)
            if (r < nn) s_nb[g * 16 + r] = abase + ld_idx_t<NW16>(p.nb_idx, ((size_t)a * K + r) * N + i);  // as an index of the staged arrays
            if (r < ns) s_ob[g * 16 + r] = (int)p.obst_idx[((size_t)a * S + r) * N + i];
        } else {
            const int cnts = (p.counts + aoff)[ii];
            nn = cnts & 0xFF; ns = cnts >> 8;
            c = (p.orient_x + aoff)[ii]; s = -(p.orient_y + aoff)[ii];  // utils.py:48-51: cos/sin of -atan2(orientation)
            const unsigned li = (unsigned)r * (unsigned)N + ii;     // inside the arena's [K][N] / [S][N] block: below 16 x 1024
            const char* nbase = (const char*)p.nb_idx + aoff * (size_t)K * (NW16 ? 2 : 1);
            if (r < nn) s_nb[g * 16 + r] = ld_idx_t<NW16>(nbase, li);  // as an index of the staged arrays (GATHER: of the arena)
            if (r < ns) s_ob[g * 16 + r] = (int)(p.obst_idx + aoff * (size_t)S)[li];
        }
    }
    s_key[g * 16 + r] = ~0ull;
    if (r == 0) { s_cnt[g] = 0; s_cnt2[g] = 0; }
    CA_OSTAMP(1);
    __syncthreads();
    CA_OSTAMP(2);

    const int M = 8 * nn + ns;
    float mx = 0.0f, my = 0.0f;
    if (M > 0) { mx = GATHER ? gpx[i] : s_px[abase + i]; my = GATHER ? gpy[i] : s_py[abase + i]; }
    if (r == 0) s_frame[g] = make_float4(c, s, mx, my);  // phase A lanes also work for the workgroup's other agents
    // ---- pre-pass: which (source, ray) pairs are worth the exact test?  Supersets only; never results. ----
    // (1) lane per agent NEIGHBOUR: all 8 octagon vertices lie on the circle of radius R around it, so the
    // rays within asin(R/d) of its direction are a superset for each of its 8 chords.  asin(t) <= t + t^3 / 6 +
    // 3 t^5 / 40 + c t^7 on [0, 1] with c = pi/2 - 1 - 1/6 - 3/40 (the series has positive coefficients and its tail
    // sums to c at t = 1): within 2e-3 rad of asin for t <= 1/2, i.e. for neighbours that do not overlap the agent.
    // The margin of 0.002 dial units = 7.8e-4 rad covers the dial's 1e-4 and the approximate reciprocal square root.
    // The pairs go to the workgroup's list at a position from the wave's prefix sum of the window widths and ONE atomic
    // per wave (at most 16 neighbours, CA_MAX_NEIGHBORS, so lane r < nn of an agent owns neighbour slot r).
    {
        const int k = r;
        int i0 = 0, w = 0;
        if (k < nn) {
            const int nb = s_nb[g * 16 + k];
            const float rx = (GATHER ? gpx[nb] : s_px[nb]) - mx, ry = (GATHER ? gpy[nb] : s_py[nb]) - my;
            s_rel[g * 16 + k] = make_float2(rx, ry);  // (env.py:288-289) for the pair trips and the winners
            const float d2 = rx * rx + ry * ry, R = p.radius;
            const float ax = c * rx - s * ry, ay = s * rx + c * ry;
            const bool all = !(d2 > 1.0404f * R * R);  // the agent is inside (or within 2 % of) that circle
            const float ua = ray_dial(ax, ay);
            const float t = R * __builtin_amdgcn_rsqf(d2);
            const float t2 = t * t;
            const float hw = t * (1.0f + t2 * (0.16666667f + t2 * (0.075f + t2 * 0.32914f))) * 2.54647908947f + 0.002f;
            int i1 = (int)floorf(ua + hw);
            i0 = (int)ceilf(ua - hw);
            if (all || i1 - i0 >= 15) { i0 = 0; i1 = 15; }
            w = max(i1 - i0 + 1, 0);
        }
        const int incl = wave_prefix_sum(w);
        int base = 0;
        if ((tid & 63) == 63) base = atomicAdd(&s_cnt[0], incl);
        base = __builtin_amdgcn_readlane(base, 63) + (incl - w);
        // (list order is irrelevant: the merge is a commutative minimum.)  A window is two to four rays wide unless the neighbour
        // overlaps the agent: four predicated stores, and a rolled loop for the rare rest -- written as one `for`, the compiler
        // builds a 16-wide vectorised loop, a 4-wide one and a remainder loop around these few stores (60 vector instructions
        // of control per wave for ~3 stores per lane)
        const unsigned short hd = (unsigned short)((g << 8) | (k << 4));
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2)
            if (t2 < w) s_pair[base + t2] = (unsigned short)(hd | ((i0 + t2) & 15));
        if (__builtin_expect(__ballot(w > 4) != 0ull, 0)) {
#pragma clang loop vectorize(disable) unroll(disable)
            for (int t2 = 4; t2 < w; ++t2) s_pair[base + t2] = (unsigned short)(hd | ((i0 + t2) & 15));
        }
    }
    // (2) lane per RAY, one obstacle edge at a time: the exact test can accept a ray only if the ray's line
    // separates the edge's end points and the crossing is not behind the origin, i.e. (up to rounding, covered
    // by tolE = 50x the error of these products) the two end points are not on the same side of the line and
    // not both behind.  Obstacle pairs fill the agent's block at the back of the list.
    {
        const float dx = s_rays[2 * r], dy = s_rays[2 * r + 1];
        int cnt2 = 0;
        for (int sidx = 0; sidx < ns; ++sidx) {
            const ObstDev o1 = load_obst(tab, s_ob[g * 16 + sidx]);
            const float x1 = o1.px - mx, y1 = o1.py - my, x2 = o1.qx - mx, y2 = o1.qy - my;
            const float ax = c * x1 - s * y1, ay = s * x1 + c * y1;
            const float bx = c * x2 - s * y2, by = s * x2 + c * y2;
            const float c2 = dx * ay - dy * ax, c3 = dx * by - dy * bx;
            const float f2 = dx * ax + dy * ay, f3 = dx * bx + dy * by;
            const float tolE = 1e-5f * p.rays[0] * (fabsf(ax) + fabsf(ay) + fabsf(bx) + fabsf(by) + 1.0f);
            const bool keep = !(c2 > tolE && c3 > tolE) && !(c2 < -tolE && c3 < -tolE) && (fmaxf(f2, f3) >= -tolE);
            const unsigned grp = (unsigned)(__ballot(keep) >> (threadIdx.x & 48)) & 0xFFFFu;  // my agent's 16 lanes
            if (keep)
                s_pair[OBS_APB * PAIRCAP - 1 - (g * SPAIRS + cnt2 + __popc(grp & ((1u << r) - 1u)))] =
                    (unsigned short)(((nn + sidx) << 4) | r);
            cnt2 += __popc(grp);
        }
        if (r == 0) s_cnt2[g] = cnt2;
    }
    CA_OSTAMP(3);
    __syncthreads();
    CA_OSTAMP(4);
    // segment m of this agent in the rotated frame (env.py:283-294, 305-315; utils.py:55-62)
    auto build = [&](int m, SegGeom& sg, float& velx, float& vely, bool want_vel) {
        float x1, y1, x2, y2, vx = 0.0f, vy = 0.0f;
        if (m < 8 * nn) {
            const int k = m >> 3, e = m & 7;
            const float2 rel = s_rel[g * 16 + k];
            const float rx = rel.x, ry = rel.y;
            const float4 oc = reinterpret_cast<const float4*>(s_oct)[e];
            x1 = oc.x + rx; y1 = oc.y + ry;
            x2 = oc.z + rx; y2 = oc.w + ry;
            if (want_vel) { const int nb = s_nb[g * 16 + k]; vx = GATHER ? gvx[nb] : s_vx[nb]; vy = GATHER ? gvy[nb] : s_vy[nb]; }  // env.py:252
        } else {
            const ObstDev o1 = load_obst(tab, s_ob[g * 16 + (m - 8 * nn)]);
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
    // chord e of a neighbour at (rx, ry) from its agent, in the agent's frame fr = (cos, sin)
    auto build_nb = [&](const float2& fr, float rx, float ry, int e, SegGeom& sg) {
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
        const float t = div_ir(sg.t_numer, denom);                     // utils.py:34 (an accepted crossing: 0 <= t <= 1, |denom| >= an ulp of O(1) products)
        hx = 0.0f + t * s10x; hy = 0.0f + t * s10y;                    // utils.py:36-37
        d = sqrt_ir(hx * hx + hy * hy);                                // utils.py:38
        return true;
    };

    // the same accept test without early exits, and the distance of an accepted crossing (utils.py:34-38)
    auto accept_nb = [&](const SegGeom& sg, float s10x, float s10y, float& denom) -> bool {
        denom = s10x * sg.s32y - sg.s32x * s10y;                      // utils.py:14
        const float s_numer = s10x * sg.s02y - s10y * sg.s02x;        // utils.py:21
        const bool dpos = denom > 0.0f;
        return (denom != 0.0f) && ((s_numer < 0.0f) != dpos) && ((sg.t_numer < 0.0f) != dpos) &&
               ((s_numer > denom) != dpos) && ((sg.t_numer > denom) != dpos);  // utils.py:15-31
    };
    auto hit_dist = [&](float t_numer, float denom, float s10x, float s10y) -> float {
        const float t = div_ir(t_numer, denom);                        // utils.py:34 (used for accepted crossings only: see hit())
        const float hx = 0.0f + t * s10x, hy = 0.0f + t * s10y;        // utils.py:36-37
        return sqrt_ir(hx * hx + hy * hy);                             // utils.py:38
    };

    // ---- phase A: lane per (source, ray) pair ----
    // An agent neighbour contributes the 8 chords of its octagon; consecutive chords share an end point
    // bit for bit (env.py:335-350 builds them as a chain), so the 8 rotated vertices are computed once
    // per pair and every chord is accept-tested against the pair's single ray.  Only accepted chords
    // (about two per pair) are re-derived through build()/hit() for the exact hit distance.
    const float tol = 2e-5f * p.rays[0] * (p.rays[0] + 2.0f * p.radius + 1.0f);  // rays[0] = neighbor_dist (env.py:321-332)
    auto merge = [&](int ga, int ray, float best, int best_m) {
        if (best_m >= 0)
            atomicMin(&s_key[ga * 16 + ray], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)best_m);
    };
    // (neighbour, ray) pairs and (obstacle edge, ray) pairs in loops of their own: a wave that mixes the two
    // kinds in one pass pays for both code paths.  The neighbour pairs of the workgroup's 16 agents form one
    // work list shared by all its lanes: an agent has 16 pairs on average in a settled crowd (C3), so a list
    // per agent would send its 16 lanes on a second trip half of the time and a list per wave (tried: 43 % of
    // the waves took a second, nearly empty trip) still wastes a third of the issue slots of this loop; with one
    // list only the first wave ever takes a second trip, for the few pairs beyond the workgroup's lane count.
    const int ntot = s_cnt[0];
    for (int pi = tid; pi < ntot; pi += OBS_BS) {
        const int pr = s_pair[pi];
        const int ga = pr >> 8, k = (pr >> 4) & 15, ray = pr & 15;
        const float2 fr = *reinterpret_cast<const float2*>(&s_frame[ga]);
        const float2 rel = s_rel[pr >> 4];  // = [ga][k]
        const float s10x = s_rays[2 * ray] - 0.0f, s10y = s_rays[2 * ray + 1] - 0.0f;
        float best = __int_as_float(0x7f800000);
        int best_m = -1;
        const float rx = rel.x, ry = rel.y;
        // Which chords can the exact test accept?  It needs the crossing parameter along the chord,
        // s_numer / denom (utils.py:21-31), inside [0, 1], i.e. the ray's LINE must separate the chord's end
        // points: with cr[e] = ray x vertex e (= -s_numer of chord e), a chord whose two end points lie on the
        // same side of the line by more than `tol` cannot be accepted.  The filter does not need the reference's
        // rounding, so it takes the cross products in the WORLD frame, where the octagon's vertices are constants:
        // ray_w x (oct_e + rel) with ray_w the ray turned back by the agent's frame -- no vertex is rotated here.
        // `tol` is 100x the rounding error of either form.  The survivors -- the entry and the exit chord, a
        // third one when the line grazes a vertex -- go through the reference's arithmetic below.
        // Vertex e of the octagon is R (cos e pi/4, -sin e pi/4) (env.py:335-350) and vertex e + 4 its mirror image,
        // so the eight cross products are four, each added to and subtracted from the ray x centre term.
        const float wx = fr.x * s10x + fr.y * s10y, wy = fr.x * s10y - fr.y * s10x;
        const float wb = wx * ry - wy * rx;
        const float R = p.radius, Rh = 0.70710678f * R;
        const float c0 = R * wy, c2 = R * wx, c1 = Rh * (wx + wy), c3 = Rh * (wy - wx);
        const float cr[8] = {wb - c0, wb - c1, wb - c2, wb + c3, wb + c0, wb + c1, wb + c2, wb - c3};
        unsigned acc = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ca = cr[e], cb = cr[(e + 1) & 7];
            const bool same_side = (ca > tol && cb > tol) || (ca < -tol && cb < -tol);
            acc |= same_side ? 0u : (1u << e);
        }
        // ascending chord index, strict '<': the first minimum wins.  Two surviving chords (entry and exit)
        // are taken together in straight-line code; the arithmetic of each is the reference's (utils.py:14-38).
        while (acc) {
            const int e1 = __ffs(acc) - 1;
            acc &= acc - 1;
            const bool two = acc != 0;
            const int e2 = two ? __ffs(acc) - 1 : e1;
            acc &= acc - 1;  // (0 & anything stays 0)
            SegGeom g1, g2;
            build_nb(fr, rx, ry, e1, g1);
            build_nb(fr, rx, ry, e2, g2);
            float dn1, dn2;
            const bool ok1 = accept_nb(g1, s10x, s10y, dn1), ok2 = accept_nb(g2, s10x, s10y, dn2) && two;
            // Both accepted (the ray enters through one chord and leaves through the other): the distance is monotone
            // in t = t_numer / denom, so when the two quotients differ by more than 1e-5 relative -- 30x what the
            // division, the products and the square root of utils.py:34-38 can round away -- the nearer chord is
            // known from a cross-multiplication and only ITS distance is computed.  Anything closer than that (the
            // ray through a shared vertex), or a crossing at the origin, takes both through the reference's
            // arithmetic and compares the rounded distances as the reference does.
            const float lhs = fabsf(g1.t_numer) * fabsf(dn2), rhs = fabsf(g2.t_numer) * fabsf(dn1);  // t1 < t2  <=>  lhs < rhs
            const bool first = ok1 && (!ok2 || lhs < rhs);
            const float tnw = first ? g1.t_numer : g2.t_numer, dnw = first ? dn1 : dn2;
            const int ew = first ? e1 : e2;
            const bool sure = fminf(lhs, rhs) < 0.99999f * fmaxf(lhs, rhs) && fmaxf(lhs, rhs) > 1e-30f &&
                              fabsf(tnw) > 1e-12f * fabsf(dnw);
            if (ok1 && ok2 && !sure) {
                const float d1 = hit_dist(g1.t_numer, dn1, s10x, s10y), d2 = hit_dist(g2.t_numer, dn2, s10x, s10y);
                if (d1 < best) { best = d1; best_m = 8 * k + e1; }
                if (d2 < best) { best = d2; best_m = 8 * k + e2; }
            } else {
                const float dw = hit_dist(tnw, dnw, s10x, s10y);
                if ((ok1 || ok2) && dw < best) { best = dw; best_m = 8 * k + ew; }
            }
        }
        merge(ga, ray, best, best_m);
    }
    const int no = s_cnt2[g];
    for (int pi = r; pi < no; pi += 16) {
        const int pr = s_pair[OBS_APB * PAIRCAP - 1 - (g * SPAIRS + pi)];
        const int k = pr >> 4, ray = pr & 15;
        const float s10x = s_rays[2 * ray] - 0.0f, s10y = s_rays[2 * ray + 1] - 0.0f;
        SegGeom sg;
        float dum0, dum1, d, hx, hy;
        const int m = 8 * nn + (k - nn);
        build(m, sg, dum0, dum1, false);
        if (hit(sg, s10x, s10y, d, hx, hy)) merge(g, ray, d, m);
    }
    CA_OSTAMP(5);
    __syncthreads();  // a ray's key takes hits from every wave of the workgroup
    CA_OSTAMP(6);
    if (!active) return;
    // ---- phase B: lane per ray ----
    const unsigned long long key = s_key[g * 16 + r];
    float bx = 0.0f, by = 0.0f, vx = 0.0f, vy = 0.0f;
    if (key != ~0ull) {
        SegGeom sg;
        float wx, wy;
        build((int)(unsigned)key, sg, wx, wy, true);
        const float s10x = s_rays[2 * r] - 0.0f, s10y = s_rays[2 * r + 1] - 0.0f;
        const float t = div_ir(sg.t_numer, s10x * sg.s32y - sg.s32x * s10y);  // utils.py:14,34: the accepted hit again
        bx = 0.0f + t * s10x; by = 0.0f + t * s10y;                      // utils.py:36-37
        if (!(bx == 0.0f && by == 0.0f)) { vx = wx; vy = wy; }  // utils.py:103
    }
    CA_OSTAMP(7);
    {   // written once, read by the caller: a non-temporal store, so that the 67 MB of a C3 step do not push the arenas'
        // state out of the L2 that the next solve reads it from
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f out = {bx, by, vx, vy};
        if constexpr (DENSE) __builtin_nontemporal_store(out, reinterpret_cast<v4f*>(p.obs) + (q * 16 + r));
        else __builtin_nontemporal_store(out, (reinterpret_cast<v4f*>(p.obs) + aoff * 16) + (ii * 16u + (unsigned)r));
    }
    CA_OSTAMP(8);
}

}  // namespace ca
