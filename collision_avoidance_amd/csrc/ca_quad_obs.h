// ca_quad_obs.h -- the laser observation inside the four-lanes-per-agent kernel (one launch per environment step)
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
//
// Why: for the small batches the quad kernel exists for, the observation as a launch of its own costs more than its
// work: C2 (1024 arenas x 16 agents) spent 11.6 of its 28.6 us per step in obs_kernel -- a launch boundary, the arena
// and the lists staged again from global memory, 4096 short waves that all sit on their SIMDs at once -- for ~4 us of
// vector issue.  Here the wave that advanced 16 agents observes for them straight away: the arena image is already
// in LDS, the neighbour lists are in its registers, the frame (orientation, position) too.
// Algorithm = obs_kernel's (ca_obs.h: conservative (source, ray) pair lists, exact test per pair, ds_min_u64 merge per
// ray, winner's velocity), with the wave's 16 agents pooled:
//   * pre-pass: lane q of a quad takes neighbours q, q + 4, ... (ray window of the octagon's circumcircle) and rays
//     q, q + 4, q + 8, q + 12 against the obstacle edges; accepted pairs go to ONE list per wave (atomic counters:
//     neighbour pairs from the front, obstacle pairs from the back; list order is irrelevant, the merge is a minimum);
//   * phase A: lane per pair over the wave's list;
//   * phase B: lane q writes rays q, q + 4, q + 8, q + 12 of its agent.
// The arithmetic of every accepted pair is the reference's (utils.py:5-40, 55-62), statement by statement the same as in
// obs_kernel, so both kernels produce the same bits (tests/test_gpu_quad.py compares either with the oracle).
#pragma once
#include "ca_obs.h"

namespace ca {

// per wave (bytes): keys [16][16] u64 | hit points [16][16] float2 | frames [16] float4 | nb slots [16][16] int |
// obstacle edges [16][4] int | counts [16] int | two counters | pair list [256 (K + S)] u16
__host__ __device__ inline size_t quad_obs_lds_per_wave(int K, int S) {
    return 2048 + 2048 + 256 + 1024 + 256 + 64 + 16 + (((size_t)256 * (K + S) * 2 + 15) & ~(size_t)15);
}

struct QuadObsWave {
    unsigned long long* key;  // [16][16]
    float2* hit;              // [16][16]
    float4* frame;            // [16] (cos, sin, pos x, pos y)
    int* nb;                  // [16][16] LDS slot of list entry k in the arena image
    int* ob;                  // [16][4] edge index into the concatenated edge tables
    int* cnt;                 // [16] nn | ns << 8
    int* counter;             // [0] neighbour pairs (list front), [1] obstacle pairs (list back)
    unsigned short* pair;     // [cap] entries (slot << 8 | source << 4 | ray)
    int cap;
};

__device__ __forceinline__ QuadObsWave quad_obs_carve(char* base, int K, int S) {
    QuadObsWave w;
    w.key = reinterpret_cast<unsigned long long*>(base);
    w.hit = reinterpret_cast<float2*>(base + 2048);
    w.frame = reinterpret_cast<float4*>(base + 4096);
    w.nb = reinterpret_cast<int*>(base + 4096 + 256);
    w.ob = reinterpret_cast<int*>(base + 4096 + 256 + 1024);
    w.cnt = reinterpret_cast<int*>(base + 4096 + 256 + 1024 + 256);
    w.counter = reinterpret_cast<int*>(base + 4096 + 256 + 1024 + 256 + 64);
    w.pair = reinterpret_cast<unsigned short*>(base + 4096 + 256 + 1024 + 256 + 64 + 16);
    w.cap = 256 * (K + S);
    return w;
}

// One wave observes for its 16 agent slots.  The caller has already left the lists of this step in the wave's tables:
// w.nb[wslot][k] = LDS slot (arena image) of agent-neighbour k, w.ob[wslot][k] = obstacle edge k as an index into `edges`.
// Arguments of the calling lane (quad lane q of wave slot wslot): whether its agent exists; nn / ns = its neighbour
// counts; (ox, oy) the orientation of its frame, pos its position; the arena image s_px .. s_vy (positions and
// velocities AFTER the step); s_tab = [32] ray end points then [32] octagon chords; out = the agent's 64-float row.
template <int KQ>
__device__ __forceinline__ void quad_obs(const QuadObsWave& w, bool active, int wslot, int q, int lane, int nn, int ns,
                                         float ox, float oy, V2 pos, const float* s_px, const float* s_py,
                                         const float* s_vx, const float* s_vy, const ObstDev* edges, const float* s_tab,
                                         float radius, float4* out) {
    const float* s_rays = s_tab;
    const float4* s_oct4 = reinterpret_cast<const float4*>(s_tab + 32);
    // ---- stage the wave's tables ----
    w.key[lane * 4 + 0] = ~0ull; w.key[lane * 4 + 1] = ~0ull; w.key[lane * 4 + 2] = ~0ull; w.key[lane * 4 + 3] = ~0ull;
    if (lane < 2) w.counter[lane] = 0;
    if (!active) { nn = 0; ns = 0; }
    const float c = ox, s = -oy;  // utils.py:48-51: cos / sin of -atan2(orientation)
    float mx = 0.0f, my = 0.0f;
    if (8 * nn + ns > 0) { mx = pos.x; my = pos.y; }
    if (q == 0) { w.frame[wslot] = make_float4(c, s, mx, my); w.cnt[wslot] = nn | (ns << 8); }
    wave_lds_sync();
    // ---- pre-pass (ca_obs.h): supersets of the (source, ray) pairs the exact test could accept; never results ----
#pragma unroll
    for (int m = 0; m < KQ; ++m) {
        const int k = 4 * m + q;
        if (k < nn) {
            const int nb = w.nb[wslot * 16 + k];
            const float rx = s_px[nb] - mx, ry = s_py[nb] - my;
            const float d2 = rx * rx + ry * ry, R = radius;
            const float ax = c * rx - s * ry, ay = s * rx + c * ry;
            const bool all = !(d2 > 1.0404f * R * R);
            const float ua = ray_dial(ax, ay);
            const float t = R * __builtin_amdgcn_rsqf(d2);
            const float hw = t * (1.0f + 0.5708f * (t * t)) * 2.54647908947f + 0.02f;
            int i0 = (int)ceilf(ua - hw), i1 = (int)floorf(ua + hw);
            if (all || i1 - i0 >= 15) { i0 = 0; i1 = 15; }
            const int wn = i1 - i0 + 1;
            if (wn > 0) {
                const int base = atomicAdd(&w.counter[0], wn);
                for (int t2 = 0; t2 < wn; ++t2)
                    w.pair[base + t2] = (unsigned short)((wslot << 8) | (k << 4) | ((i0 + t2) & 15));
            }
        }
    }
    const float nd = s_rays[0];  // = neighbor_dist (env.py:321-332)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int r = 4 * r4 + q;
        const float dx = s_rays[2 * r], dy = s_rays[2 * r + 1];
        for (int sidx = 0; sidx < ns; ++sidx) {
            const ObstDev o1 = load_obst(edges, w.ob[wslot * 4 + sidx]);
            const float x1 = o1.px - mx, y1 = o1.py - my, x2 = o1.qx - mx, y2 = o1.qy - my;
            const float ax = c * x1 - s * y1, ay = s * x1 + c * y1;
            const float bx = c * x2 - s * y2, by = s * x2 + c * y2;
            const float c2 = dx * ay - dy * ax, c3 = dx * by - dy * bx;
            const float f2 = dx * ax + dy * ay, f3 = dx * bx + dy * by;
            const float tolE = 1e-5f * nd * (fabsf(ax) + fabsf(ay) + fabsf(bx) + fabsf(by) + 1.0f);
            const bool keep = !(c2 > tolE && c3 > tolE) && !(c2 < -tolE && c3 < -tolE) && (fmaxf(f2, f3) >= -tolE);
            if (keep) {
                const int at = atomicAdd(&w.counter[1], 1);
                w.pair[w.cap - 1 - at] = (unsigned short)((wslot << 8) | (sidx << 4) | r);
            }
        }
    }
    wave_lds_sync();

    const float tol = 2e-5f * nd * (nd + 2.0f * radius + 1.0f);
    auto merge = [&](int ga, int ray, float best, int best_m, float bhx, float bhy) {
        if (best_m >= 0) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)best_m;
            atomicMin(&w.key[ga * 16 + ray], key);
            if (w.key[ga * 16 + ray] == key) w.hit[ga * 16 + ray] = make_float2(bhx, bhy);
        }
    };
    // chord e of a neighbour at (rx, ry) relative to the agent, in the agent's frame fr = (cos, sin, x, y) (utils.py:55-62)
    auto build_nb = [&](const float4& fr, float rx, float ry, int e, SegGeom& sg) {
        const float4 oc = s_oct4[e];
        const float x1 = oc.x + rx, y1 = oc.y + ry, x2 = oc.z + rx, y2 = oc.w + ry;
        sg.r1x = fr.x * x1 - fr.y * y1; sg.r1y = fr.y * x1 + fr.x * y1;  // utils.py:59
        sg.r2x = fr.x * x2 - fr.y * y2; sg.r2y = fr.y * x2 + fr.x * y2;  // utils.py:60
        sg.s32x = sg.r2x - sg.r1x; sg.s32y = sg.r2y - sg.r1y;
        sg.s02x = 0.0f - sg.r1x; sg.s02y = 0.0f - sg.r1y;
        sg.t_numer = sg.s32x * sg.s02y - sg.s32y * sg.s02x;
    };
    // utils.py:5-40 for the ray with end point (s10x, s10y) starting at the origin, without early exits (ca_obs.h hit_nb)
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
    // ---- phase A: lane per (neighbour, ray) pair of the wave's 16 agents ----
    const int ntot = w.counter[0];
    for (int pi = lane; pi < ntot; pi += 64) {
        const int pr = w.pair[pi];
        const int ga = pr >> 8, k = (pr >> 4) & 15, ray = pr & 15;
        const float4 fr = w.frame[ga];
        const float s10x = s_rays[2 * ray] - 0.0f, s10y = s_rays[2 * ray + 1] - 0.0f;
        float best = __int_as_float(0x7f800000), bhx = 0.0f, bhy = 0.0f;
        int best_m = -1;
        const int nb = w.nb[ga * 16 + k];
        const float rx = s_px[nb] - fr.z, ry = s_py[nb] - fr.w;
        // which chords can the exact test accept: those whose end points the ray's line separates (ca_obs.h)
        const float wx = fr.x * s10x + fr.y * s10y, wy = fr.x * s10y - fr.y * s10x;
        const float wb = wx * ry - wy * rx;
        float cr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 oc = s_oct4[e];
            cr[e] = (wx * oc.y - wy * oc.x) + wb;
        }
        unsigned acc = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ca = cr[e], cb = cr[(e + 1) & 7];
            const bool same_side = (ca > tol && cb > tol) || (ca < -tol && cb < -tol);
            acc |= same_side ? 0u : (1u << e);
        }
        while (acc) {  // ascending chord index, strict '<': the first minimum wins
            const int e1 = __ffs(acc) - 1;
            acc &= acc - 1;
            const bool two = acc != 0;
            const int e2 = two ? __ffs(acc) - 1 : e1;
            acc &= acc - 1;
            SegGeom g1, g2;
            build_nb(fr, rx, ry, e1, g1);
            build_nb(fr, rx, ry, e2, g2);
            float d1, d2, h1x, h1y, h2x, h2y;
            const bool ok1 = hit_nb(g1, s10x, s10y, d1, h1x, h1y), ok2 = hit_nb(g2, s10x, s10y, d2, h2x, h2y) && two;
            if (ok1 && d1 < best) { best = d1; best_m = 8 * k + e1; bhx = h1x; bhy = h1y; }
            if (ok2 && d2 < best) { best = d2; best_m = 8 * k + e2; bhx = h2x; bhy = h2y; }
        }
        merge(ga, ray, best, best_m, bhx, bhy);
    }
    // ---- phase A: lane per (obstacle edge, ray) pair ----
    const int nto = w.counter[1];
    for (int pi = lane; pi < nto; pi += 64) {
        const int pr = w.pair[w.cap - 1 - pi];
        const int ga = pr >> 8, sidx = (pr >> 4) & 15, ray = pr & 15;
        const float4 fr = w.frame[ga];
        const int nng = w.cnt[ga] & 0xFF;
        const float s10x = s_rays[2 * ray] - 0.0f, s10y = s_rays[2 * ray + 1] - 0.0f;
        const ObstDev o1 = load_obst(edges, w.ob[ga * 4 + sidx]);
        const float x1 = o1.px - fr.z, y1 = o1.py - fr.w, x2 = o1.qx - fr.z, y2 = o1.qy - fr.w;
        SegGeom sg;
        sg.r1x = fr.x * x1 - fr.y * y1; sg.r1y = fr.y * x1 + fr.x * y1;  // utils.py:59
        sg.r2x = fr.x * x2 - fr.y * y2; sg.r2y = fr.y * x2 + fr.x * y2;  // utils.py:60
        sg.s32x = sg.r2x - sg.r1x; sg.s32y = sg.r2y - sg.r1y;
        sg.s02x = 0.0f - sg.r1x; sg.s02y = 0.0f - sg.r1y;
        sg.t_numer = sg.s32x * sg.s02y - sg.s32y * sg.s02x;
        // utils.py:5-40 with its early exits (ca_obs.h hit)
        const float denom = s10x * sg.s32y - sg.s32x * s10y;          // utils.py:14
        bool ok = denom != 0.0f;
        const bool dpos = denom > 0.0f;
        const float s_numer = s10x * sg.s02y - s10y * sg.s02x;        // utils.py:21
        ok = ok && !((s_numer < 0.0f) == dpos) && !((sg.t_numer < 0.0f) == dpos) &&
             !(((s_numer > denom) == dpos) || ((sg.t_numer > denom) == dpos));
        if (ok) {
            const float t = sg.t_numer / denom;                        // utils.py:34
            const float hx = 0.0f + t * s10x, hy = 0.0f + t * s10y;    // utils.py:36-37
            const float d = sqrtf(hx * hx + hy * hy);                  // utils.py:38
            merge(ga, ray, d, 8 * nng + sidx, hx, hy);
        }
    }
    wave_lds_sync();
    if (!active) return;
    // ---- phase B: lane q writes rays q, q + 4, q + 8, q + 12 of its agent ----
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int r = 4 * r4 + q;
        const unsigned long long key = w.key[wslot * 16 + r];
        float bx = 0.0f, by = 0.0f, vx = 0.0f, vy = 0.0f;
        if (key != ~0ull) {
            const int m = (int)(unsigned)key;
            float x1, y1, ovx = 0.0f, ovy = 0.0f;
            if (m < 8 * nn) {  // env.py:283-294: chord e of neighbour k, moving with the neighbour (env.py:252)
                const int k = m >> 3, e = m & 7;
                const int nb = w.nb[wslot * 16 + k];
                const float rx = s_px[nb] - mx, ry = s_py[nb] - my;
                const float4 oc = s_oct4[e];
                x1 = oc.x + rx; y1 = oc.y + ry;
                ovx = s_vx[nb]; ovy = s_vy[nb];
            } else {           // env.py:305-315: a static obstacle edge
                const ObstDev o1 = load_obst(edges, w.ob[wslot * 4 + (m - 8 * nn)]);
                x1 = o1.px - mx; y1 = o1.py - my;
            }
            const float r1x = c * x1 - s * y1, r1y = s * x1 + c * y1;      // utils.py:59
            const float lvx = x1 + ovx, lvy = y1 + ovy;                      // utils.py:57
            const float rvx = c * lvx - s * lvy, rvy = s * lvx + c * lvy;    // utils.py:61
            const float wvx = rvx - r1x, wvy = rvy - r1y;                    // utils.py:62
            const float2 h = w.hit[wslot * 16 + r];
            bx = h.x; by = h.y;
            if (!(bx == 0.0f && by == 0.0f)) { vx = wvx; vy = wvy; }       // utils.py:103
        }
        out[r] = make_float4(bx, by, vx, vy);
    }
}

}  // namespace ca
