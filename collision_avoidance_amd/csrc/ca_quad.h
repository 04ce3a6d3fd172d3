// ca_quad.h -- the solve kernel with FOUR LANES PER AGENT, and T steps per launch
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
//
// Why: with one lane per agent a batch of small arenas is a few hundred waves -- BASELINE config C2 (1024 arenas x 16
// agents) is 256 waves on a chip of 1024 SIMDs, and each of them is one long dependent chain (a lone wave issues an
// instruction every 5-10 cycles).  Here the quad of lanes 4 s .. 4 s + 3 works for agent slot s, so the same batch is
// four times as many waves and every phase of an agent's step is dealt over its quad:
//   * obstacle edges and neighbour candidates are visited four at a time, each lane keeping a sorted list of 64-bit
//     (distance, index) keys; the four lists are merged by two quad-permute exchanges (DPP register moves) and a bitonic
//     merge network of v_min_f64 / v_max_f64: the keys are totally ordered (the index breaks ties exactly as the
//     contract's visiting order does), so the merged list IS the serial scan's list;
//   * the obstacle half-planes are built one per lane; the contract's "already covered by an earlier obstacle line"
//     test only decides whether a line is DROPPED, so every lane tests its edge against the lines of the lanes before it
//     and the drops are resolved in list order with three quad broadcasts;
//   * the agent half-planes are built one per lane and round; all lines go to the quad's slot of the wave's LDS table
//     in the contract's order;
//   * LP2 walks the lines in order (every lane of the quad holds the same running result), each LP1 inside it deals
//     its clips to the four lanes and merges tLeft (a maximum), tRight (a minimum) and the failure flag (an OR) -- all
//     independent of the order, so the values are those of the serial loop bit for bit (ca_lp.h lp3_coop, whose code
//     solves LP3 here too: the quad's slot already holds its lines);
//   * the per-agent scalar work (fp64 goal direction, done test, RNG) is done redundantly by the four lanes.
// The kernel advances p.T steps in one launch when no action tensor and no observation is asked for (ca_rollout:
// env.py:570-573 / ALAN:106-123, the ORCA-only loops): an arena never leaves its workgroup, so position, velocity,
// preferred velocity, target and the done flag stay in registers between steps, the arena image in LDS, and the only
// global traffic inside the loop is the rare event (re-goal, arrival, reset).  Neighbour lists are written by the last
// step of the launch (every step under CA_F_FREEZE, where an arena's last active step is not known in advance).
// Supported: N <= 128 agents per arena (64 with K = 10 and more than four obstacle neighbours), K <= 10, up to 16 obstacle
// neighbours (two instantiations: 4 and 16); ca_env.hip picks it where one lane per agent would leave SIMDs without a
// wave (CA_QUAD=0/1 forces it).
#pragma once
#include "ca_step.h"

namespace ca {

template <int J>
__device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xF, 0xF, false); }
template <int J>
__device__ __forceinline__ float quad_bcast(float v) { return __int_as_float(quad_bcast<J>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double quad_xor_key(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = quad_xor<CTRL>((int)(unsigned)b), hi = quad_xor<CTRL>((int)(unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// sum over the quad (every lane gets it)
__device__ __forceinline__ int quad_sum(int v) {
    v += quad_xor<0xB1>(v);
    v += quad_xor<0x4E>(v);
    return v;
}

// sorted insert into the first NN slots of an ascending key list (see ca_nbr.h sorted_insert)
template <int NN, int LEN>
__device__ __forceinline__ void sorted_insert_n(double (&key)[LEN], double x) {
    static_for<NN - 1>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = NN - 1 - decltype(kc)::value;  // NN-1 .. 1
        key[k] = key_max(key[k - 1], key_min(key[k], x));
    });
    key[0] = key_min(key[0], x);
}

// Two ascending lists of M keys (mine, the partner lane's) -> the M smallest of their union, ascending, in both lanes:
// c[k] = min(mine[k], other[M-1-k]) is a bitonic sequence holding exactly those keys; log2(M) half-cleaner stages sort it.
// (every index a compile-time constant: the lists live in registers)
template <int M, int H>
__device__ __forceinline__ void half_cleaner(double (&key)[M]) {
    if constexpr (H >= 1) {
        static_for<M>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr ((k & H) == 0) {
                const double lo = key_min(key[k], key[k + H]), hi = key_max(key[k], key[k + H]);
                key[k] = lo; key[k + H] = hi;
            }
        });
        half_cleaner<M, H / 2>(key);
    }
}
template <int M, int CTRL>
__device__ __forceinline__ void merge_with_partner(double (&key)[M]) {
    double c[M];
    static_for<M>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; c[k] = quad_xor_key<CTRL>(key[k]); });
    static_for<M>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; key[k] = key_min(key[k], c[M - 1 - k]); });
    half_cleaner<M, M / 2>(key);
}
template <int M>
__device__ __forceinline__ void merge_quad(double (&key)[M]) {
    merge_with_partner<M, 0xB1>(key);
    merge_with_partner<M, 0x4E>(key);
}

// index part of entry k of a register list, k = M4 + q with M4 a compile-time constant and q the lane's position in its
// quad.  (The candidates pass through opaque moves: left visible, the compiler turns the select chain into ONE indexed
// load and, for that, keeps the whole list in scratch memory.)
template <int M4, int M>
__device__ __forceinline__ int pick4_index(const double (&key)[M], int q) {
    int v = key_index(key[M4]);
    asm volatile("" : "+v"(v));
    if constexpr (M4 + 1 < M) { int w = key_index(key[M4 + 1]); asm volatile("" : "+v"(w)); v = (q == 1) ? w : v; }
    if constexpr (M4 + 2 < M) { int w = key_index(key[M4 + 2]); asm volatile("" : "+v"(w)); v = (q == 2) ? w : v; }
    if constexpr (M4 + 3 < M) { int w = key_index(key[M4 + 3]); asm volatile("" : "+v"(w)); v = (q == 3) ? w : v; }
    return v;
}

// App. A.5 LP2 (dirOpt = false) for the quad's slot of the wave's line table: `n` lines in the contract's order at
// ls.get(0 .. n-1).  Every lane of the quad holds the same `result`; returns the index of the first infeasible line or n.
__device__ __forceinline__ int lp2_quad(const LdsLines& ls, int n, int q, float radius, V2 opt, V2& result) {
    if (absSq(opt) > sqr(radius)) result = normalize_ir(opt) * radius;   // (|opt| > radius here)
    else result = opt;
    int fail = n;
    Line L = ls.get(0);  // (row 0 exists in LDS whatever n is)
    for (int i = 0; i < n; ++i) {
        const Line Li = L;
        if (i + 1 < n) L = ls.get(i + 1);  // the next line is in flight while this one is tested
        if (det(Li.dir, Li.point - result) > 0.0f) {
            const float dp = dot(Li.point, Li.dir);
            const float disc = sqr(dp) + sqr(radius) - absSq(Li.point);
            int failed = disc < 0.0f ? 1 : 0;
            const float sq = sqrt_ir(disc);
            float tLeft = -dp - sq;
            float tRight = -dp + sq;
            for (int jj = q; jj < i; jj += 4) {
                const Line M = ls.get(jj);
                const float den = det(Li.dir, M.dir);
                const float num = det(M.dir, Li.point - M.point);
                const bool par = fabsf(den) <= EPS;
                const float t = div_ir(num, den);   // (ca_lp.h lp1_reg: in range wherever it is used)
                const bool right = !par && den >= 0.0f, left = !par && !(den >= 0.0f);
                tRight = (right && t < tRight) ? t : tRight;
                tLeft = (left && tLeft < t) ? t : tLeft;
                failed |= (par && num < 0.0f) ? 1 : 0;
            }
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
            if (failed || tLeft > tRight) {  // LP1 fails: LP2 stops with the result it had
                fail = i;
                break;
            }
            const float t = dot(Li.dir, opt - Li.point);
            if (t < tLeft) result = Li.point + tLeft * Li.dir;
            else if (t > tRight) result = Li.point + tRight * Li.dir;
            else result = Li.point + t * Li.dir;
        }
    }
    return fail;
}

// LDS of the quad kernel (bytes): line table [waves][2 ML][16] float4 | px py vx vy [BS/4] | per-arena reductions
// [BS/4][4] int | rewards [BS/4] double
// (+ ALAN instantiation: weights | times | softmax terms, [n_actions][BS/4] doubles each)
__host__ __device__ inline size_t quad_lds_bytes(int BS, int KMAX, int SQ, int n_actions = 0) {
    const size_t ns = (size_t)BS / 4;
    return (size_t)(BS / 64) * (2 * (SQ + KMAX)) * POOL_SLOTS * 16 + ns * 16 + ns * 16 + ns * 8 + 3 * ns * (size_t)n_actions * 8;
}

// SQ: obstacle-neighbour capacity of the variant (S <= SQ): 4 (the synthetic crowds: one boundary polygon) or 16 (the
// reference's own worlds: doorway, blocks, tube -- env.py:77-123, ALAN:175-457)
// ALAN: the online bandit of ALAN_true.py:569-628 around every step of the launch (ca_alan.h has the same arithmetic as kernels
// of their own for the lane-per-agent path): softmax over the agent's action weights -> one draw -> preferred velocity = goal
// direction rotated by the action -> [the ORCA step] -> reward of the action -> sliding-window update of weights and times.
// Weights and times live in LDS for the whole launch ([action][agent slot] fp64), the four lanes of a quad share the
// actions (exp64 of the softmax, the window update), the draw is keyed by (seed, global arena, agent, episode, step) as ever.
template <int KMAX, int BS, int SQ, bool ALAN = false>
__global__ __launch_bounds__(BS) void quad_kernel(const StepArgs p) {
    static_assert(POOL_SLOTS == 16, "a wave holds 16 quads: one line-table slot each");
    static_assert(SQ == 4 || SQ == 16, "obstacle lists of 4 or 16");
    static_assert(!CA_NBW16(BS / 4), "the quad kernel stores 8-bit agent-neighbour ids: at most 256 agent slots per workgroup "
                                     "(and not the CA_VAR_NB16 diagnostic build)");
    constexpr int M = KMAX <= 4 ? 4 : (KMAX <= 8 ? 8 : 16);  // merge width of the agent-neighbour lists
    constexpr int ML = SQ + KMAX;
    constexpr int NS = BS / 4;                               // agent slots per workgroup
    constexpr int KQ = (KMAX + 3) / 4;                       // agent-line rounds
    extern __shared__ float4 smem4[];
    const ColdK& c = *(ColdK*)p.cold;
    const int tid = threadIdx.x;
    const int q = tid & 3, slot = tid >> 2;
    const int P = p.P;
    const int la = slot >> p.logP;
    const int i = slot & (P - 1);
    const int apb = NS >> p.logP;
    const int a = p.a0 + (int)blockIdx.x * apb + la;
    const int N = p.N, K = p.K, S = p.S;
    const bool in_arena = (a < p.a1) && (i < N);
    const int gq = in_arena ? a * N + i : 0;
    const int lbase = la << p.logP;

    float4* pool = smem4 + (size_t)(tid >> 6) * (2 * ML) * POOL_SLOTS;
    float4* hdr = pool + (size_t)(2 * ML - 1) * POOL_SLOTS;
    const int wslot = (tid & 63) >> 2;  // this quad's slot of its wave's line table
    LdsLines ls; ls.base = pool + wslot; ls.stride = POOL_SLOTS;
    float* s_px = reinterpret_cast<float*>(smem4 + (size_t)(BS / 64) * (2 * ML) * POOL_SLOTS);
    float* s_py = s_px + NS;
    float* s_vx = s_py + NS;
    float* s_vy = s_vx + NS;
    int* s_red = reinterpret_cast<int*>(s_vy + NS);            // [NS][4]; arena la uses row la
    double* s_rew = reinterpret_cast<double*>(s_red + NS * 4);  // [NS]
    double* s_w = s_rew + NS;                                   // ALAN: [nA][NS] weights | [nA][NS] times | [nA][NS] softmax terms
    int* red = s_red + la * 4;  // per arena: [0] not-done agents, [1] pairs, [2] wall hits, [3] goals

    // ---- state of this agent, resident for the whole launch (the four lanes of a quad hold the same values) ----
    V2 pos = mk(0.0f, 0.0f), vel = mk(0.0f, 0.0f), pref = mk(0.0f, 0.0f);
    double gx = 0.0, gy = 0.0;
    int done = 1, steps = 0, adone = 0, epi = 0, rc = 0;
    bool touched = false;  // some step of this launch advanced the agent (a frozen arena is left exactly as it is)
    if (in_arena) {
        pos = mk(p.pos_x[gq], p.pos_y[gq]);
        vel = mk(p.vel_x[gq], p.vel_y[gq]);
        pref = mk(p.pref_x[gq], p.pref_y[gq]);
        gx = c.goal_x[gq]; gy = c.goal_y[gq];
        done = c.agent_done[gq];
        steps = c.step_count[a];
        adone = p.arena_done[a];
        epi = c.episode[a];
        rc = c.regoal_count[gq];
    }
    const int tab0 = (p.tab_off != nullptr && in_arena) ? p.tab_off[a] : 0;
    const int nedge = (p.tab_off != nullptr) ? (in_arena ? p.tab_off[a + 1] - tab0 : 0) : p.n_obst;
    const ObstDev* tab = p.obst + tab0;  // this arena's edge table
    const float R = p.radius;
    const bool nodone = (p.flags & 8u) != 0;  // CA_F_NODONE
    // per-arena counters of the launch (meaningful in the lanes of agent 0), flushed once at the end
    unsigned acc_coll = 0, acc_wall = 0, acc_goals = 0, acc_epis = 0, acc_frozen = 0, acc_ovf = 0, acc_steps = 0;
    unsigned long long lastep = 0;
    bool have_lastep = false;
    float ox = pref.x, oy = pref.y;
    const double KEY_EMPTY = __longlong_as_double(0x7F800000FFFFFFFFll);  // (+inf, -1)

    typedef const __attribute__((address_space(4))) AlanCold AlanK;
    int nA = 0, last_id = 0;
    float last_rew = 0.0f;
    double acc_rew = 0.0;
    if constexpr (ALAN) {
        const AlanK& al = *(AlanK*)p.alan;
        nA = al.nA;
        for (int k = q; k < nA; k += 4) {
            s_w[k * NS + slot] = in_arena ? al.w[((size_t)a * nA + k) * N + i] : 0.0;
            s_w[(nA + k) * NS + slot] = in_arena ? al.t[((size_t)a * nA + k) * N + i] : 0.0;
        }
        wave_lds_sync();
    }
    double* s_t = s_w + nA * NS;
    double* s_ps = s_t + nA * NS;

    const int T = p.actions ? 1 : (p.T > 0 ? p.T : 1);
    for (int t = 0; t < T; ++t) {
        const bool frozen = (p.flags & 16u) != 0 && in_arena && adone != 0;  // CA_F_FREEZE: the episode of this arena is over
        const bool active = in_arena && !frozen;
        if (frozen && i == 0) acc_frozen += 1;
        if (active && i == 0) acc_steps += 1;
        touched = touched || active;
        const bool write_lists = (t == T - 1) || (p.flags & 16u) != 0;

        CA_STAMP(0);
#if defined(CA_STAMPS) && CA_STAMPS == 3
        CA_STAMP_HWID();   // placement diagnostic: slots 2, 3 = HW_ID, XCC_ID (the phase stamps 2, 3 are skipped below)
#endif
        // ---- preferred velocity of this step (env.py:371-383) and the arena image ----
        V2 pf32 = mk(1.0f, 0.0f);
        if (active && p.actions) {
            double pf_x, pf_y, sn, cs;
            pref_dir64(pos.x, pos.y, gx, gy, &pf_x, &pf_y);
            sincos64((double)p.actions[gq], &sn, &cs);
            const double rl_x = pf_x * cs - pf_y * sn;
            const double rl_y = pf_x * sn + pf_y * cs;
            pf32 = mk((float)pf_x, (float)pf_y);
            pref = mk((float)rl_x, (float)rl_y);
        }
        int act_id = 0;
        double dgx = 1.0, dgy = 0.0, dlx = 1.0, dly = 0.0;   // goal direction and rotated direction of this step (ALAN:588-595)
        if constexpr (ALAN) {
            const AlanK& al = *(AlanK*)p.alan;
            if (active)
                for (int k = q; k < nA; k += 4) s_ps[k * NS + slot] = exp64(s_w[k * NS + slot] / al.temp);   // ALAN:580-581
            wave_lds_sync();
            if (active) {   // (the four lanes alike from here: every lane of the quad holds the draw and the directions)
                const double sum = np_sum(nA, [&](int k) { return s_ps[k * NS + slot]; });
                double acc = 0.0;
                for (int k = 0; k < nA; ++k) {   // ALAN:582: the normalised terms, in order (the four lanes of the quad store the
                    const double v = s_ps[k * NS + slot] / sum;   // same value to the same word: one instruction, in lockstep)
                    s_ps[k * NS + slot] = v;
                    acc += v;
                }
                double ui, u1;
                if (p.alan_u) ui = p.alan_u[gq];
                else rng2(c.seed, c.arena_offset + a, i, RNG_ALAN + (epi << 8), (uint32_t)steps, &ui, &u1);
                act_id = nA - 1;   // np.random.choice (ALAN:585): first action whose normalised cdf exceeds u
                double run = 0.0;
                bool found = false;
                for (int k = 0; k < nA - 1; ++k) {
                    run += s_ps[k * NS + slot];
                    if (!found && run / acc > ui) { act_id = k; found = true; }
                }
                pref_dir64(pos.x, pos.y, gx, gy, &dgx, &dgy);                       // ALAN:588
                const double cs = al.act_c[act_id], sn = al.act_s[act_id];         // ALAN:592-595
                dlx = dgx * cs - dgy * sn; dly = dgx * sn + dgy * cs;
                pref = mk((float)dlx, (float)dly);                                  // ALAN:598
            }
        }
        if (q == 0) { s_px[slot] = pos.x; s_py[slot] = pos.y; s_vx[slot] = vel.x; s_vy[slot] = vel.y; }
        __syncthreads();

        CA_STAMP(1);
        // ---- obstacle neighbours (App. A.2): edges e = q, q + 4, ... ----
        double okey[SQ];
#pragma unroll
        for (int k = 0; k < SQ; ++k) okey[k] = KEY_EMPTY;
        int oin = 0;
        {
            const float rangeSq = sqr(p.time_horizon_obst * p.max_speed + p.radius);
            for (int e0 = 0; __ballot(e0 < nedge) != 0ull; e0 += 4) {
                const int e = e0 + q;
                const bool mine = active && e < nedge;
                const ObstDev o1 = load_obst(tab, mine ? e : 0);
                const V2 a1 = mk(o1.px, o1.py), a2 = mk(o1.qx, o1.qy);
                const float alol = leftOf(a1, a2, pos);
                const float dsl = div_ir(sqr(alol), absSq(a2 - a1));   // (an edge has a length; the quotient is only compared with the range)
                const float dsq = distSqPointSegment(a1, a2, pos);
                const bool in = mine && dsl < rangeSq && alol < 0.0f && dsq < rangeSq;
                oin += in ? 1 : 0;
                sorted_insert_n<SQ>(okey, in ? make_key(dsq, e) : KEY_EMPTY);
            }
        }
        merge_quad<SQ>(okey);
        oin = quad_sum(oin);
        const int ocnt = oin < S ? oin : S;
        if (__builtin_expect(oin > S && q == 0, 0)) { acc_ovf += 1; note_overflow(p.cold, a, i, oin); }

#if !defined(CA_STAMPS) || CA_STAMPS != 3
        CA_STAMP(2);
#endif
        // ---- agent neighbours (App. A.2): candidates j = q, q + 4, ... ----
        double nkey[M];
#pragma unroll
        for (int k = 0; k < M; ++k) nkey[k] = KEY_EMPTY;
        if (K > 0) {
            const float rangeSq0 = sqr(p.neighbor_dist);
            const int trips = (N + 3) >> 2;
            V2 o_next = mk(s_px[lbase + q], s_py[lbase + q]);  // (slot lbase + q exists: P >= ... see the host check)
            for (int tr = 0; tr < trips; ++tr) {
                const int j = 4 * tr + q;
                const V2 o = o_next;
                if (tr + 1 < trips) { const int jn = (j + 4 < P) ? j + 4 : 0; o_next = mk(s_px[lbase + jn], s_py[lbase + jn]); }
                const float dsq = absSq(pos - o);
                const bool ok = active && j < N && j != i && dsq < rangeSq0;
                sorted_insert_n<KMAX>(nkey, ok ? make_key(dsq, j) : KEY_EMPTY);
            }
            merge_quad<M>(nkey);
        }
        int ncnt = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) ncnt += (k < K && key_index(nkey[k]) >= 0) ? 1 : 0;

        // ---- the lists are state (the reference's reset() observes with the lists of the last doStep) ----
        if (active && write_lists) {
            // (indices behind an opaque move: the K + S store addresses are formed per step -- hoisted out of the T-step loop they
            // are a dozen 64-bit values that the 512-lane shapes spill and reload every step)
            int a_l = a, i_l = i;
            if constexpr (BS >= 512) asm volatile("" : "+v"(a_l), "+v"(i_l));   // (the smaller shapes have the registers: 1-2 % faster hoisted)
            if (q == 0) p.counts[a_l * N + i_l] = (unsigned short)(ncnt | (ocnt << 8));
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if ((k & 3) == q && k < K) st_idx_t<false>(p.nb_idx, ((size_t)a_l * K + k) * N + i_l, key_index(nkey[k]));
#pragma unroll
            for (int k = 0; k < SQ; ++k)
                if ((k & 3) == q && k < S) p.obst_idx[((size_t)a_l * S + k) * N + i_l] = (unsigned short)key_index(okey[k]);
        }

#if !defined(CA_STAMPS) || CA_STAMPS != 3
        CA_STAMP(3);
#endif
        // ---- obstacle ORCA lines (App. A.3): in round r lane q builds the line of obstacle neighbour 4 r + q ----
        int no = 0;
        {
            const float invTO = 1.0f / p.time_horizon_obst;
            const float thr = invTO * R;
            static_for<SQ / 4>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                if (__ballot(4 * r < ocnt) != 0ull) {  // some quad of the wave has a neighbour in this round
                    const int e = pick4_index<4 * r>(okey, q);
                    const bool have = active && 4 * r + q < ocnt;
                    float lpx = 0.0f, lpy = 0.0f, ldx = 1.0f, ldy = 0.0f;
                    V2 c1 = mk(0.0f, 0.0f), c2 = mk(0.0f, 0.0f);
                    bool ex = false;
                    if (have) {
                        const ObstDev E = load_obst(tab, e);
                        c1 = invTO * (mk(E.px, E.py) - pos);
                        c2 = invTO * (mk(E.qx, E.qy) - pos);
                        ex = obst_orca_line4(tab, e, pos, vel, R, invTO, [](V2, V2) { return false; }, lpx, lpy, ldx, ldy);
                    }
                    // "already covered" (App. A.3 step 1): an edge covered by an existing earlier line yields no line at
                    // all.  Earlier lines = those of the rounds before (in the quad's table already) ...
                    int alive = have ? 1 : 0;  // not skipped so far
                    if constexpr (r > 0) {
                        for (int j = 0; j < no; ++j) {
                            const Line Mj = ls.get(j);
                            if (det(c1 - Mj.point, Mj.dir) - thr >= -EPS && det(c2 - Mj.point, Mj.dir) - thr >= -EPS) alive = 0;
                        }
                    }
                    // ... and those of the lanes before this one in this round, resolved in list order
                    bool cov0, cov1, cov2;
                    {
                        const V2 pt = mk(quad_bcast<0>(lpx), quad_bcast<0>(lpy)), dr = mk(quad_bcast<0>(ldx), quad_bcast<0>(ldy));
                        cov0 = det(c1 - pt, dr) - thr >= -EPS && det(c2 - pt, dr) - thr >= -EPS;
                    }
                    {
                        const V2 pt = mk(quad_bcast<1>(lpx), quad_bcast<1>(lpy)), dr = mk(quad_bcast<1>(ldx), quad_bcast<1>(ldy));
                        cov1 = det(c1 - pt, dr) - thr >= -EPS && det(c2 - pt, dr) - thr >= -EPS;
                    }
                    {
                        const V2 pt = mk(quad_bcast<2>(lpx), quad_bcast<2>(lpy)), dr = mk(quad_bcast<2>(ldx), quad_bcast<2>(ldy));
                        cov2 = det(c1 - pt, dr) - thr >= -EPS && det(c2 - pt, dr) - thr >= -EPS;
                    }
                    const int exi = (have && ex) ? 1 : 0;  // would produce a line if it is not skipped
                    // line J of the round exists iff it was not skipped and produced a line
                    { const int e0 = quad_bcast<0>(exi & alive); if (q > 0 && e0 && cov0) alive = 0; }
                    { const int e1 = quad_bcast<1>(exi & alive); if (q > 1 && e1 && cov1) alive = 0; }
                    { const int e2 = quad_bcast<2>(exi & alive); if (q > 2 && e2 && cov2) alive = 0; }
                    const bool exists = (exi & alive) != 0;
                    const unsigned qm = (unsigned)(__ballot(exists) >> (tid & 60)) & 0xFu;
                    if (exists) ls.put(no + __popc(qm & ((1u << q) - 1u)), Line{mk(lpx, lpy), mk(ldx, ldy)});
                    no += __popc(qm);
                    if constexpr (r + 1 < SQ / 4) wave_lds_sync();  // the next round reads these rows
                }
            });
        }
        CA_STAMP(4);
        // ---- agent ORCA lines (App. A.4): lane q builds the lines of neighbours q, q + 4, ... ----
        {
            const float invT = 1.0f / p.time_horizon;
            const float invDt = 1.0f / p.time_step;
            static_for<KQ>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                const int k = 4 * m + q;
                const int j = pick4_index<4 * m>(nkey, q);
                if (active && k < ncnt) {
                    const int sj = lbase + j;
                    ls.put(no + k, agent_orca_line(pos, vel, mk(s_px[sj], s_py[sj]), mk(s_vx[sj], s_vy[sj]), R, invT, invDt));
                }
            });
        }
        wave_lds_sync();
        CA_STAMP(5);
        // ---- 2-D linear program (App. A.5) ----
        const int nl = active ? no + ncnt : 0;
        V2 nv = mk(0.0f, 0.0f);
        const int fail = lp2_quad(ls, nl, q, p.max_speed, pref, nv);
        CA_STAMP(6);
        {   // LP3 for the quads whose LP2 was infeasible: their slot already holds the lines
            const bool need = fail < nl;
            if (__ballot(need) != 0ull) {
                if (q == 0) hdr[wslot] = make_float4(nv.x, nv.y, __int_as_float(need ? (nl | (no << 8) | (fail << 16)) : 0), 0.0f);
                wave_lds_sync();
                lp3_coop(pool, ML, POOL_SLOTS, p.max_speed);
                wave_lds_sync();
                if (need) { const float4 h = hdr[wslot]; nv = mk(h.x, h.y); }
            }
        }
        if (active) {  // ---- integrate (App. A.1) ----
            vel = nv;
            pos = pos + vel * p.time_step;
        }
        CA_STAMP(7);
        // ---- epilogue (ca_step.h, same order of operations) ----
        __syncthreads();  // every lane is done with the pre-step arena image
        if (q == 0) { s_px[slot] = pos.x; s_py[slot] = pos.y; }
        if (q == 0) { s_red[slot * 4 + 0] = 0; s_red[slot * 4 + 1] = 0; s_red[slot * 4 + 2] = 0; s_red[slot * 4 + 3] = 0; }
        __syncthreads();

        if (p.flags & 2u) {  // CA_F_STATS: overlapping pairs (i < j), agents touching a wall; dealt over the quad
            int pairs = 0;
            const float crSq = sqr(R + R);
            if (active)
                for (int j = i + 1 + q; j < N; j += 4)
                    if (absSq(pos - mk(s_px[lbase + j], s_py[lbase + j])) < crSq) ++pairs;
            bool wall = false;
            if (active)
                for (int e = q; e < nedge; e += 4) {
                    const ObstDev o1 = load_obst(tab, e);
                    if (distSqPointSegment(mk(o1.px, o1.py), mk(o1.qx, o1.qy), pos) < sqr(R)) wall = true;
                }
            pairs = quad_sum(pairs);
            const int walls = quad_sum(wall ? 1 : 0);
            if (active && q == 0) {
                if (pairs) atomicAdd(&red[1], pairs);
                if (walls) atomicAdd(&red[2], 1);
            }
        }
        CA_STAMP(8);
        // ---- reward (env.py:389-400) or preferred velocity towards the goal (env.py:449) ----
        float rew = 0.0f;
        if (active) {
            if (p.actions) {
                const float scale = (float)c.reward_scale;
                const float r_goal = vel.x * pf32.x + vel.y * pf32.y;
                const float r_polite = vel.x * pref.x + vel.y * pref.y;
                rew = scale * r_goal + (1.0f - scale) * r_polite;
                if (q == 0) c.reward[gq] = rew;
            } else {
                double dx, dy;
                pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
                pref = mk((float)dx, (float)dy);
            }
        }
        CA_STAMP(9);
        // ---- step counter and done test (env.py:352-365, 404-410; ALAN:118-121, 547-566) ----
        bool goal_changed = false;
        if (active && !p.actions && !nodone) ++steps;
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
                // (the element index behind an opaque move: the addresses of these rare stores are formed here, per event --
                // hoisted out of the T-step loop they are nine 64-bit values carried, and at 512 lanes spilled, through every step)
                int gq_e = gq;
                if constexpr (BS >= 512) asm volatile("" : "+v"(gq_e));
                if (c.done_mode == 2) {
                    double u0, u1;
                    rng2(c.seed, c.arena_offset + a, i, RNG_REGOAL, (uint32_t)rc, &u0, &u1);
                    gx = uniform64((double)c.goal_x0, (double)c.goal_x1, u0);
                    gy = uniform64((double)c.goal_y0, (double)c.goal_y1, u1);
                    rc += 1;
                    if (q == 0) c.regoal_count[gq_e] = rc;
                } else {
                    done = 1;
                    gx = c.goal2_x[gq_e]; gy = c.goal2_y[gq_e];
                    if (q == 0) { c.arrive_step[gq_e] = steps; c.agent_done[gq_e] = 1; }
                }
                if (q == 0) { c.goal_x[gq_e] = gx; c.goal_y[gq_e] = gy; }
                goal_changed = true;
                if (q == 0) atomicAdd(&red[3], 1);
            }
        }
        if (active && p.actions) ++steps;
        if (active && done == 0 && q == 0) atomicAdd(&red[0], 1);
        __syncthreads();

        bool all_done = false;
        if (active) {
            all_done = !nodone && (red[0] == 0);
            if (c.max_step > 0 && steps >= c.max_step) all_done = true;
        }
        const bool do_reset = all_done && (p.flags & 4u);  // CA_F_AUTORESET
        if (do_reset) {  // env.py:461-488 for this arena
            double u0, u1;
            rng2(c.seed, c.arena_offset + a, i, RNG_RESET, (uint32_t)epi, &u0, &u1);
            pos = mk((float)uniform64((double)c.spawn_x0, (double)c.spawn_x1, u0),
                     (float)uniform64((double)c.spawn_y0, (double)c.spawn_y1, u1));
            done = 0;
            int gq_r = gq;
            if constexpr (BS >= 512) asm volatile("" : "+v"(gq_r));
            if (q == 0) c.agent_done[gq_r] = 0;
            double dx, dy;
            pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
            pref = mk((float)dx, (float)dy);
        }
        CA_STAMP(10);
        // sum of rewards: the fixed-shape tree of ca_step.h over agent slots (one lane per slot here)
        if (p.actions && (p.flags & 2u)) {
            if (q == 0) s_rew[slot] = active ? (double)rew : 0.0;
            __syncthreads();
            if (tid < NS) {
                double r = s_rew[tid];
                const int w = P < 64 ? P : 64;
                for (int off = w >> 1; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
                int tid2 = tid;   // (opaque at 512 lanes: the addresses below are derived per step, not hoisted out of the T-step loop
                if constexpr (BS >= 512) asm volatile("" : "+v"(tid2));   // and carried -- spilled -- through every step of it)
                const int la2 = tid2 >> p.logP, i2 = tid2 & (P - 1), a2 = p.a0 + (int)blockIdx.x * apb + la2;
                if (a2 < p.a1 && i2 < N && (i2 & 63) == 0 && !((p.flags & 16u) != 0 && p.arena_done[a2] != 0))
                    atomicAdd(reinterpret_cast<double*>(&c.arena_stats[(size_t)a2 * ST_STRIDE + ST_SUMREW]), r);
            }
        }
        // orientation of the observation frame (env.py:236): direction to the goal from the final state
        if (active) {
            ox = pref.x; oy = pref.y;
            if (!do_reset && (p.actions != nullptr || goal_changed)) {
                double dx, dy;
                pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
                ox = (float)dx; oy = (float)dy;
            }
        }
        if constexpr (ALAN) {   // reward of the executed action and the sliding-window update (ALAN:603-628)
            const AlanK& al = *(AlanK*)p.alan;
            if (active) {
                {   // env.py:389-400 in fp32, as ca_step reports it
                    const float scale = (float)al.reward_scale;
                    const float r_goal = vel.x * (float)dgx + vel.y * (float)dgy;
                    const float r_polite = vel.x * (float)dlx + vel.y * (float)dly;
                    last_rew = scale * r_goal + (1.0f - scale) * r_polite;
                    acc_rew += (double)last_rew;
                }
                const double vx = (double)vel.x, vy = (double)vel.y;
                const double Rw = al.reward_scale * (vx * dgx + vy * dgy) + (1.0 - al.reward_scale) * (vx * dlx + vy * dly);
                for (int k = q; k < nA; k += 4) {
                    double tk = s_t[k * NS + slot] + al.dt;
                    double wk = s_w[k * NS + slot];
                    if (tk >= al.window) { tk = 0.0; wk = 0.0; }
                    if (k == act_id) wk = Rw;
                    s_t[k * NS + slot] = tk; s_w[k * NS + slot] = wk;
                }
                pref = mk((float)dlx, (float)dly);   // the agent still holds the velocity it was given at ALAN:598
                last_id = act_id;
            }
            wave_lds_sync();
        }
        __syncthreads();  // all lanes have read red[] and episode[]
        if (active && i == 0 && q == 0) {
            acc_coll += (unsigned)red[1]; acc_wall += (unsigned)red[2]; acc_goals += (unsigned)red[3];
            if (all_done) {
                acc_epis += 1;
                lastep = ((unsigned long long)(unsigned)steps << 32) | (unsigned)(N - red[0]);
                have_lastep = true;
            }
            if (do_reset) c.episode[a] = epi + 1;
        }
        if (active) {
            adone = all_done ? 1 : 0;
            if (do_reset) { steps = 0; epi += 1; }
        }
        // (the next step's first barrier separates these reads of red[] from its clearing)
        CA_STAMP(11);
    }

    // ---- write the state back, once ----
    if constexpr (ALAN) {
        const AlanK& al = *(AlanK*)p.alan;
        if (in_arena && touched) {
            for (int k = q; k < nA; k += 4) {
                al.w[((size_t)a * nA + k) * N + i] = s_w[k * NS + slot];
                al.t[((size_t)a * nA + k) * N + i] = s_t[k * NS + slot];
            }
            if (q == 0) {
                al.action[gq] = last_id;
                al.reward[gq] = last_rew;
                if (p.flags & 2u) atomicAdd(reinterpret_cast<double*>(&c.arena_stats[(size_t)a * ST_STRIDE + ST_SUMREW]), acc_rew);
            }
        }
    }
    if (in_arena && q == 0 && acc_frozen && i == 0) c.arena_stats[(size_t)a * ST_STRIDE + ST_FROZEN] += acc_frozen;
    if (in_arena && q == 0 && touched) {
        c.orient_x[gq] = ox; c.orient_y[gq] = oy;
        c.pos_x[gq] = pos.x; c.pos_y[gq] = pos.y;
        c.vel_x[gq] = vel.x; c.vel_y[gq] = vel.y;
        c.pref_x[gq] = pref.x; c.pref_y[gq] = pref.y;
        if (acc_ovf) atomicAdd(reinterpret_cast<unsigned*>(&p.arena_stats[(size_t)a * ST_STRIDE + ST_OVERFLOW]), acc_ovf);
        if (i == 0) {
            unsigned long long* st = c.arena_stats + (size_t)a * ST_STRIDE;
            if (acc_coll) st[ST_COLL] += acc_coll;
            if (acc_wall) st[ST_OBST_COLL] += acc_wall;
            if (acc_goals) st[ST_GOALS] += acc_goals;
            if (acc_epis) st[ST_EPISODES] += acc_epis;
            if (have_lastep) st[ST_LASTEP] = lastep;
            c.arena_done[a] = adone;
            c.step_count[a] = steps;
            atomicAdd(&c.arena_steps[a], (unsigned long long)acc_steps);
        }
    }
}

}  // namespace ca
