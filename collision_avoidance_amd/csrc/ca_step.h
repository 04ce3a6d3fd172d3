// ca_step.h -- the solve kernel (neighbour search, ORCA lines, LP, integration, reward / done) and the reset kernels
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
#pragma once
#include "ca_lp.h"
#include "ca_lines.h"
#include "ca_nbr.h"
#include "ca_alan.h"

namespace ca {

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
typedef const __attribute__((address_space(4))) StepCold ColdK;  // the cold block through the constant address space

// An agent with MORE obstacle neighbours than the register-line kernel has obstacle slots (ST): solved apart, by its own
// lane alone, exactly as the contract says -- every half-plane built in order into an LDS table of its own (the wave's LP3
// pool is free at that point: it holds TS such tables side by side, column g = table g, so up to TS such agents of a wave are
// solved at the same time), LP2 over it; the caller runs LP3 on the same tables with lp3_coop (stride TS) where LP2 fails.
// RVO2 keeps every edge in range (env.py:249, 301-318), so the list capacity stays 16; in the reference's doorway and
// "congested" worlds (14 edges) no agent-step of 3.8e5 sampled had more than four (profiles/r04_b_reference_worlds.txt),
// which is what lets those worlds run on the register-line kernel.  Not inlined: the hot path's registers are not its.
// (the pointers cross the call boundary WITH their address spaces -- LDS for the table and the staged arena, global for the edge
// records and the lists: as generic pointers every access in here was a FLAT instruction)
#define CA_AS(n) __attribute__((address_space(n)))
template <bool NW16, int TS>
__device__ __noinline__ void solve_many_obstacles(CA_AS(3) char* tbl3, int MLX, const CA_AS(1) char* tab1, const CA_AS(1) char* oidx1,
                                                 const CA_AS(1) char* nidx1, int stride, int ocnt, int ncnt, const CA_AS(3) char* arena3, int bs,
                                                 V2 pos, V2 vel, V2 pref, float R, float invTO, float invT, float invDt, float max_speed) {
    // (few enough arguments to travel in registers: one on the stack would give the whole kernel a scratch segment)
    float4* tbl = (float4*)tbl3;
    const ObstDev* tab = (const ObstDev*)tab1;
    const unsigned short* oidx = (const unsigned short*)oidx1;
    const void* nidx = (const void*)nidx1;
    const float* arena = (const float*)arena3;
    const float *ax = arena, *ay = arena + bs, *avx = arena + 2 * bs, *avy = arena + 3 * bs;  // the staged arena: px | py | vx | vy
    LdsLines ls; ls.base = tbl; ls.stride = TS;
    int nl = 0;
    for (int s = 0; s < ocnt; ++s) {
        const int e = ld_idx_t<true>(oidx, (size_t)s * (size_t)stride);
        Line line;
        auto covered = [&](V2 c1, V2 c2) {
            for (int j = 0; j < nl; ++j) {
                const Line M = ls.get(j);
                if (det(c1 - M.point, M.dir) - invTO * R >= -EPS && det(c2 - M.point, M.dir) - invTO * R >= -EPS) return true;
            }
            return false;
        };
        if (obst_orca_line(tab, e, pos, vel, R, invTO, covered, line)) { ls.put(nl, line); ++nl; }
    }
    const int numObst = nl;
    for (int k = 0; k < ncnt; ++k) {
        const int j = ld_idx_t<NW16>(nidx, (size_t)k * (size_t)stride);
        ls.put(nl, agent_orca_line(pos, vel, mk(ax[j], ay[j]), mk(avx[j], avy[j]), R, invT, invDt));
        ++nl;
    }
    V2 nv = mk(0.0f, 0.0f);
    const int fail = lp2(ls, nl, max_speed, pref, false, nv);
    // lp3_coop's slot header; the caller reads the result (and whether LP3 is needed: fail < nl) from it
    tbl[(2 * MLX - 1) * TS] = make_float4(nv.x, nv.y, __int_as_float(nl | (numObst << 8) | (fail << 16)), 0.0f);
}

#ifndef CA_LB512
#define CA_LB512 4   // waves per SIMD the 512-lane register-line kernel is built for (diagnostic: 2 = 256 VGPRs)
#endif
// HELP = 2: launched with 2 BS lanes, the upper half helps in the neighbour scan and ends (ca_nbr.h)
// SMX: capacity of the obstacle-neighbour list (S <= SMX).  SMX > ST (register lines): the rare agent with more than ST
// obstacle neighbours is solved apart (solve_many_obstacles)
// ALAN (kernels of one and two waves, K <= 10 -- register lines with obstacle lists of 4 or 16, and the LDS line table: every world
// of the reference's ALAN runs, ALAN:738-772): the online bandit of ALAN_true.py:569-628 around the step, as in the
// four-lanes kernel (ca_quad.h) -- softmax draw and rotated preferred velocity in the prologue (the softmax terms wait in the wave's
// LP3 pool, which is free then: at most ML actions), reward and the sliding-window update of weights / times (global memory,
// [A][nA][N]) where the epilogue begins; the goal direction is derived again there from the staged pre-step position instead of
// living in registers across the solve.  Same arithmetic as ca_alan.h's kernels, which remain the three-launch form of the rest.
template <int KMAX, int BS, int ST, bool FUSE, int HELP = 1, int SMX = (ST > 0 ? ST : SMAX), bool ALAN = false>
__global__ __launch_bounds__(BS * HELP, ST > 0 ? (BS == 512 ? CA_LB512 : 4) : 1) void step_kernel(const StepArgs p) {
    extern __shared__ float4 smem4[];
    constexpr bool LISTP = BS > 64;   // the pair count of the statistics goes through the neighbour lists (arenas within one wave: the
    //                                   scan of the staged arena is as fast -- round 5 measured the lists there: 57.7 against 57.3 us)
    __shared__ unsigned s_vmax2;   // (LISTP) the largest squared speed of the workgroup's arenas in this step, as float bits: see the pair count
    if constexpr (LISTP) { if (threadIdx.x == 0) s_vmax2 = 0u; }   // (barriers follow before its first use)
    CA_PRIO_START();
    // FUSE: the neighbour search runs at the head of this kernel instead of in a launch of its own (one
    // drain/fill less per step, and its dispatch skew overlaps useful work).  A lane later reads back only the
    // lists of its own agent, which it wrote itself; the search's LDS arrays are not used again.
    if constexpr (FUSE) { if (nbr_body<KMAX, BS, SMX, HELP>(p)) return; }
    constexpr int ML = ST + KMAX;  // register slots (ST > 0)
    const int tid = threadIdx.x;
    const int P = p.P;
    int la, i;
    lane_slot(p, tid, la, i);
    const int apb = p.apb;
    const int a = p.a0 + work_block(p) * apb + la;
    const bool frozen = arena_frozen(p, a);  // CA_F_FREEZE: the episode of this arena is over
    const bool active = (a < p.a1) && (i < p.N) && (la < apb) && !frozen;
    if (frozen && i == 0) p.arena_stats[(size_t)a * ST_STRIDE + ST_FROZEN] += 1;
    const int N = p.N, K = p.K, S = p.S;
    const int q = active ? a * N + i : 0;
    const int lbase = tid - i;   // the arena's first lane (= its first slot of the staged arrays)

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
    // Long-lived per-agent values are kept OUT of registers across the solve (the register-line LP runs at the
    // 128-VGPR limit of four waves per SIMD, and what the allocator spills goes to scratch = HBM traffic): the
    // goal direction waits in the lane's s_misc slot, the position is re-read from the staged arena, the done flag
    // is loaded where the done test needs it.
    // goal direction and action-rotated direction of env.py:375-381, rounded to fp32 as the reward uses them (env.py:394-399
    // on the simulator's floats); the fp64 targets are read again where the done test needs them, instead of living
    // in four registers across the solve
    V2 pf32 = mk(1.0f, 0.0f);
    if (active) {
        pos = mk(p.pos_x[q], p.pos_y[q]);
        vel = mk(p.vel_x[q], p.vel_y[q]);
        if (p.actions) {  // env.py:371-383
            double pf_x, pf_y, sn, cs;
            pref_dir64(pos.x, pos.y, p.goal_x[q], p.goal_y[q], &pf_x, &pf_y);
            sincos64((double)p.actions[q], &sn, &cs);
            const double rl_x = pf_x * cs - pf_y * sn;
            const double rl_y = pf_x * sn + pf_y * cs;
            pf32 = mk((float)pf_x, (float)pf_y);
            pref = mk((float)rl_x, (float)rl_y);
        } else if constexpr (ALAN) {   // ALAN:578-598: softmax over the action weights, one draw, goal direction rotated by the action
            typedef const __attribute__((address_space(4))) AlanCold AlanK;
            const AlanK& al = *(AlanK*)p.alan;
            const int nA = al.nA;
            // [k][lane] in the wave's LP3 pool (ML doubles per lane), or -- LDS line table -- over the table itself, which is
            // empty until the barrier below (2 (K + S) doubles per lane); ca_alan_configure checks that the actions fit
            constexpr int PSTR = ST > 0 ? 64 : BS;
            double* ps = ST > 0 ? reinterpret_cast<double*>(s_lines + (size_t)(tid >> 6) * (2 * ML) * POOL_SLOTS) + (tid & 63)
                                : reinterpret_cast<double*>(s_lines) + tid;
            const double* w = al.w + (size_t)a * nA * N + i;
            for (int k = 0; k < nA; ++k) ps[k * PSTR] = exp64(w[(size_t)k * N] / al.temp);
            const double sum = np_sum(nA, [&](int k) { return ps[k * PSTR]; });
            double acc = 0.0;
            for (int k = 0; k < nA; ++k) { const double v = ps[k * PSTR] / sum; ps[k * PSTR] = v; acc += v; }
            double ui, u1;
            if (p.alan_u) ui = p.alan_u[q];
            else {
                const StepCold* cp = p.cold;
                rng2(cp->seed, cp->arena_offset + a, i, RNG_ALAN + (cp->episode[a] << 8), (uint32_t)cp->step_count[a], &ui, &u1);
            }
            int act_id = nA - 1;
            double run = 0.0;
            bool found = false;
            for (int k = 0; k < nA - 1; ++k) {
                run += ps[k * PSTR];
                if (!found && run / acc > ui) { act_id = k; found = true; }
            }
            double dgx, dgy;
            pref_dir64(pos.x, pos.y, p.goal_x[q], p.goal_y[q], &dgx, &dgy);
            const double cs = al.act_c[act_id], sn = al.act_s[act_id];
            pref = mk((float)(dgx * cs - dgy * sn), (float)(dgx * sn + dgy * cs));
            pf32 = mk(__int_as_float(act_id), 0.0f);   // (the lane's slot of the goal direction carries the action: no action tensor here)
        } else {
            pref = mk(p.pref_x[q], p.pref_y[q]);
        }
    }
    s_px[tid] = pos.x; s_py[tid] = pos.y; s_vx[tid] = vel.x; s_vy[tid] = vel.y;
    reinterpret_cast<float*>(s_misc)[tid * 4 + 0] = pf32.x; reinterpret_cast<float*>(s_misc)[tid * 4 + 1] = pf32.y;
    // workgroups of more than one wave carry more per-lane state: there the preferred velocity waits in the lane's LDS
    // slot too (it is needed by every LP1 and by the epilogue; kept in registers it was spilled to scratch memory and
    // reloaded 14 times inside LP2 -- tools/kernel_resources.py)
    constexpr bool PARK_PREF = ST > 0 && (BS > 64 || SMX > ST);
    if constexpr (PARK_PREF) { reinterpret_cast<float*>(s_misc)[tid * 4 + 2] = pref.x; reinterpret_cast<float*>(s_misc)[tid * 4 + 3] = pref.y; }
    __syncthreads();

    CA_STAMP(1);
    CA_PRIO_POINT(1);
    // ---- neighbour lists of this step (App. A.2), produced by nbr_kernel ----
    // (the list pointers as scalars of their own: left inside the 16-register tuple their kernel-argument load
    // arrives in, every use after a spill reloads the whole tuple -- 16 v_readlane for one pointer)
    const void* nb_idx_s = p.nb_idx;
    const unsigned short* obst_idx_s = p.obst_idx;
    asm volatile("" : "+s"(nb_idx_s), "+s"(obst_idx_s));
    const int cnts = active ? (int)p.counts[q] : 0;
    const int ocnt = cnts >> 8, ncnt = cnts & 0xFF;
    const ObstDev* tab = p.obst + ((p.tab_off != nullptr && active) ? p.tab_off[a] : 0);  // this arena's edge table
#if defined(CA_STAMPS) && CA_STAMPS == 3
    CA_STAMP_HWID();
#else
    CA_STAMP(2);
    CA_STAMP(3);
#endif
    const float R = p.radius;
    V2 nv = mk(0.0f, 0.0f);
    if constexpr (ST > 0) {
        // ================= register path =================
        float4 L[ML];
        static_for<ML>([&](auto kc) __attribute__((always_inline)) { L[decltype(kc)::value] = make_float4(0.0f, 0.0f, 1.0f, 0.0f); });
        int no = 0;  // obstacle lines produced so far (slots [0, no))
        // more obstacle neighbours than line slots (SMX > ST only): this lane sits the register path out and is solved below
        const bool many = (SMX > ST) && active && ocnt > ST;
        const bool fast = active && !many;
        {
            const float invTO = 1.0f / p.time_horizon_obst;
            int e_next = (fast && ocnt > 0) ? ld_idx_t<true>(obst_idx_s, ((size_t)a * S + 0) * N + i) : 0;
            const int s_end = S < ST ? S : ST;
            for (int s = 0; s < s_end; ++s) {
                if (fast && s < ocnt) {
                    const int e = e_next;
                    if (s + 1 < ocnt) e_next = ld_idx_t<true>(obst_idx_s, ((size_t)a * S + (s + 1)) * N + i);
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
                    if (obst_orca_line(tab, e, pos, vel, R, invTO, covered, line)) {
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
#if defined(CA_STAMPS) && CA_STAMPS == 3
        if ((tid & 63) == 0 && p.dbg) p.dbg[((size_t)blockIdx.x * (BS / 64) + (tid >> 6)) * 16 + 4] = 0;  // LP3 rounds | lanes << 32
#else
        CA_STAMP(4);
#endif
        CA_PRIO_POINT(2);
        {
            const float invT = 1.0f / p.time_horizon;
            const float invDt = 1.0f / p.time_step;
            int jn[KMAX];  // all neighbour indices in flight at once
            static_for<KMAX>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                jn[k] = (k < ncnt) ? ld_idx_t<CA_NBW16(BS)>(nb_idx_s, ((size_t)a * K + k) * N + i) : 0;
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
        CA_PRIO_POINT(3);
        // ---- 2-D linear program (App. A.5) on the register slots ----
        const int nl = no + ncnt;
        int fail = nl;
        auto opt_fn = [&]() __attribute__((always_inline)) -> V2 {
            if constexpr (PARK_PREF) return mk(reinterpret_cast<const float*>(s_misc)[tid * 4 + 2], reinterpret_cast<const float*>(s_misc)[tid * 4 + 3]);
            else return pref;
        };
        if (fast) fail = lp2_reg<ML, ST>(L, no, ncnt, p.max_speed, opt_fn, nv);
        CA_STAMP(6);
        CA_PRIO_POINT(5);
        // ---- LP3 for the lanes whose LP2 was infeasible: they copy their lines into a slot of the
        // wave's LDS pool and solve there; more than POOL_SLOTS such lanes take further rounds ----
        {
            float4* pool = s_lines + (size_t)(tid >> 6) * (2 * ML) * POOL_SLOTS;
            float4* hdr = pool + (size_t)(2 * ML - 1) * POOL_SLOTS;
            bool need = active && fail < nl;
            const unsigned long long below = (1ull << (tid & 63)) - 1ull;
            while (true) {
                const unsigned long long m = __ballot(need);
#if defined(CA_STAMPS) && CA_STAMPS == 3
                if ((tid & 63) == 0 && p.dbg) p.dbg[((size_t)blockIdx.x * (BS / 64) + (tid >> 6)) * 16 + 4] += 1 + ((unsigned long long)__popcll(m) << 32);
#endif
                if (__builtin_expect(!m, 1)) break;
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
            if constexpr (SMX > ST) {   // the agents with more than ST obstacle neighbours, GX at a time (rare in small worlds; every
                // sixth agent-step of the two-way tube, which takes this kernel when its batch is large: ca_env.hip pick_variant)
                // (everything this stage needs is derived afresh from the lane id, the staged arena and the lists the lane
                // wrote itself -- behind an opaque move, so that nothing of it lives in registers across the solve)
                constexpr int MLX = SMX + KMAX;
                constexpr int GX = (2 * ML * POOL_SLOTS) / (2 * MLX);   // tables of 2 MLX rows side by side in the wave's pool
                static_assert(GX >= 1 && GX <= POOL_SLOTS, "the many-obstacle tables fit the wave's pool");
                int tid_o = threadIdx.x;
                asm volatile("" : "+v"(tid_o));
                int la_o, i_o;
                lane_slot(p, tid_o, la_o, i_o);
                const int a_o = p.a0 + work_block(p) * apb + la_o;
                const bool act_o = (a_o < p.a1) && (i_o < N) && (la_o < apb) && !arena_frozen(p, a_o);
                const int cnts_o = act_o ? (int)p.counts[a_o * N + i_o] : 0;
                bool need_o = (cnts_o >> 8) > ST;
                const unsigned long long below_o = (1ull << (tid_o & 63)) - 1ull;
                // (the constants of the call below, read again behind opaque moves: shared with the hot path's copies they would
                // be carried through the whole solve in vector registers for the sake of this rare stage -- and spilled)
                float rad_o = p.radius, tho_o = p.time_horizon_obst, th_o = p.time_horizon, dt_o = p.time_step, ms_o = p.max_speed;
                asm volatile("" : "+s"(rad_o), "+s"(tho_o), "+s"(th_o), "+s"(dt_o), "+s"(ms_o));
                while (true) {
                    const unsigned long long om = __ballot(need_o);
                    if (__builtin_expect(!om, 1)) break;     // (rare: tells the register allocator that this loop is cold)
                    const int rank = __popcll(om & below_o);
                    const bool mine = need_o && rank < GX;
                    if (mine) {
                        const ObstDev* tab_o = p.obst + (p.tab_off != nullptr ? p.tab_off[a_o] : 0);
                        solve_many_obstacles<CA_NBW16(BS), GX>((CA_AS(3) char*)(pool + rank), MLX, (const CA_AS(1) char*)tab_o,
                                                               (const CA_AS(1) char*)(obst_idx_s + (size_t)a_o * S * N + i_o),
                                                               (const CA_AS(1) char*)nb_idx_s + ((size_t)a_o * K * N + i_o) * (CA_NBW16(BS) ? 2 : 1), N,
                                                               cnts_o >> 8, cnts_o & 0xFF, (const CA_AS(3) char*)(s_px + (tid_o - i_o)), BS,
                                                               mk(s_px[tid_o], s_py[tid_o]), mk(s_vx[tid_o], s_vy[tid_o]), opt_fn(), rad_o,
                                                               1.0f / tho_o, 1.0f / th_o, 1.0f / dt_o, ms_o);
                    }
                    wave_lds_sync();
                    const int waiting = __popcll(om);
                    const int nt = waiting < GX ? waiting : GX;    // tables of this round
                    bool lp3 = false;
                    if ((tid_o & 63) < nt) {
                        const int hz = __float_as_int(pool[(2 * MLX - 1) * GX + (tid_o & 63)].z);   // n | numObst << 8 | fail << 16
                        lp3 = ((hz >> 16) & 0xFF) < (hz & 0xFF);
                    }
                    if (__ballot(lp3) != 0ull) {   // LP2 failed somewhere: LP3 on the same tables, four lanes each (a table whose LP2
                        lp3_coop(pool, MLX, nt, ms_o, GX);   // succeeded has begin = n: untouched)
                        wave_lds_sync();
                    }
                    if (mine) { const float4 h = pool[(2 * MLX - 1) * GX + rank]; nv = mk(h.x, h.y); need_o = false; }
                    wave_lds_sync();   // (the next round rewrites the tables)
                }
            }
        }
    } else {
        // ================= LDS-table path =================
        int nl = 0;
    {
        const float invTO = 1.0f / p.time_horizon_obst;
        int e_next = (ocnt > 0) ? ld_idx_t<true>(obst_idx_s, ((size_t)a * S + 0) * N + i) : 0;
        for (int s = 0; s < S; ++s) {
            if (s < ocnt) {
                const int e = e_next;
                if (s + 1 < ocnt) e_next = ld_idx_t<true>(obst_idx_s, ((size_t)a * S + (s + 1)) * N + i);
                Line line;
                auto covered = [&](V2 c1, V2 c2) {
                    for (int j = 0; j < nl; ++j) {
                        const Line M = ls.get(j);
                        if (det(c1 - M.point, M.dir) - invTO * R >= -EPS && det(c2 - M.point, M.dir) - invTO * R >= -EPS)
                            return true;
                    }
                    return false;
                };
                if (obst_orca_line(tab, e, pos, vel, R, invTO, covered, line)) {
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
        int j_next = (ncnt > 0) ? ld_idx_t<CA_NBW16(BS)>(nb_idx_s, ((size_t)a * K + 0) * N + i) : 0;
        for (int k = 0; k < K; ++k) {
            if (k < ncnt) {
                const int j = lbase + j_next;
                if (k + 1 < ncnt) j_next = ld_idx_t<CA_NBW16(BS)>(nb_idx_s, ((size_t)a * K + (k + 1)) * N + i);
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
    if (active && fail < nl) lp3<KMAX + SMAX>((__attribute__((address_space(3))) char*)ls.base, ls.stride, nl, numObstLines, fail, p.max_speed, nv);
    }
    if (active) {  // ---- integrate (App. A.1) ----
        vel = nv;
        int tid_i = threadIdx.x;  // (behind an opaque move: the LDS address is derived here, not carried across the solve)
        asm volatile("" : "+v"(tid_i));
        pos = mk(s_px[tid_i], s_py[tid_i]) + vel * p.time_step;  // own pre-step position: still in the staged arena
    }
    CA_STAMP(7);
    CA_PRIO_POINT(6);
    // ---- epilogue.  Its indices are derived afresh from the lane id (behind an opaque move, so that the compiler
    // cannot keep the prologue's copies -- 64-bit element offsets, LDS addresses -- alive across the solve, where
    // they would be spilled to scratch); the names shadow the prologue's on purpose. ----
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const StepCold* cold_e = p.cold;
    asm volatile("" : "+s"(cold_e));  // the loads through it stay behind this point
    {
    // (constant address space: uniform scalar loads; after the opaque move the compiler no longer knows where the
    // pointer came from and would use per-lane flat loads)
    const ColdK& c = *(ColdK*)cold_e;
    const int tid = tid_e;
    int la, i;
    lane_slot(p, tid, la, i);
    const int a = p.a0 + work_block(p) * apb + la;
    const int q = active ? a * N + i : 0;
    const int lbase = tid - i;
    const ObstDev* tab = p.obst + ((p.tab_off != nullptr && active) ? p.tab_off[a] : 0);
    pf32 = mk(reinterpret_cast<float*>(s_misc)[tid * 4 + 0], reinterpret_cast<float*>(s_misc)[tid * 4 + 1]);
    if constexpr (PARK_PREF) pref = mk(reinterpret_cast<float*>(s_misc)[tid * 4 + 2], reinterpret_cast<float*>(s_misc)[tid * 4 + 3]);
    float rew_alan = 0.0f;
    if constexpr (ALAN) {   // ALAN:603-628: reward of the executed action, sliding-window update (the staged arena still is the pre-step one)
        typedef const __attribute__((address_space(4))) AlanCold AlanK;
        const AlanK& al = *(AlanK*)p.alan;
        if (active) {
            const int act_id = __float_as_int(pf32.x), nA = al.nA;
            double dgx, dgy;
            pref_dir64(s_px[tid], s_py[tid], c.goal_x[q], c.goal_y[q], &dgx, &dgy);   // the prologue's values again, bit for bit
            const double cs = al.act_c[act_id], sn = al.act_s[act_id];
            const double dlx = dgx * cs - dgy * sn, dly = dgx * sn + dgy * cs;
            {
                const float scale = (float)al.reward_scale;
                const float r_goal = vel.x * (float)dgx + vel.y * (float)dgy;
                const float r_polite = vel.x * (float)dlx + vel.y * (float)dly;
                rew_alan = scale * r_goal + (1.0f - scale) * r_polite;
                al.reward[q] = rew_alan;
            }
            const double vx = (double)vel.x, vy = (double)vel.y;
            const double Rw = al.reward_scale * (vx * dgx + vy * dgy) + (1.0 - al.reward_scale) * (vx * dlx + vy * dly);
            double* w = al.w + (size_t)a * nA * N + i;
            double* t = al.t + (size_t)a * nA * N + i;
            for (int k = 0; k < nA; ++k) {
                double tk = t[(size_t)k * N] + al.dt;
                double wk = w[(size_t)k * N];
                if (tk >= al.window) { tk = 0.0; wk = 0.0; }
                if (k == act_id) wk = Rw;
                t[(size_t)k * N] = tk; w[(size_t)k * N] = wk;
            }
            al.action[q] = act_id;
            c.pref_x[q] = (float)dlx; c.pref_y[q] = (float)dly;   // the agent still holds the velocity it was given at ALAN:598
        }
    }

    if constexpr (LISTP) {   // how far does any agent of the arena move in this step?
        const unsigned sp = wave_max_u32(active ? __float_as_uint(absSq(vel)) : 0u);
        if ((tid & 63) == 63) atomicMax(&s_vmax2, sp);
    }
    __syncthreads();  // every lane is done with the pre-step arena image
    // the post-step positions, staged for the pair count below.  Arenas within one wave (!LISTP): as (x, y) pairs, TWICE per
    // arena -- entries [2 lbase + i] and [2 lbase + N + i] of the float2 view of the four staged arrays --, so that the balanced
    // scan reads "agent (i + d) mod N" as entry i + d without a wrap
    float2* s_xy2 = reinterpret_cast<float2*>(s_px);
    if constexpr (LISTP) { s_px[tid] = pos.x; s_py[tid] = pos.y; }
    else if (i < p.N && la < apb) { s_xy2[2 * lbase + i] = make_float2(pos.x, pos.y); s_xy2[2 * lbase + i + p.N] = make_float2(pos.x, pos.y); }
    s_misc[tid * 4 + 0] = 0; s_misc[tid * 4 + 1] = 0; s_misc[tid * 4 + 2] = 0; s_misc[tid * 4 + 3] = 0;
    __syncthreads();
    int* red = s_misc + la * 4;  // per-arena: [0] not-done agents, [1] pairs, [2] wall hits, [3] goals

    if (p.flags & 2u) {  // CA_F_STATS
        // Overlapping pairs (i < j, distance < 2R after the step).  Nobody moves farther than m = the arena's largest speed of
        // this step x dt (measured, not assumed: two agents that a reset drops onto the same spot can leave the linear programs
        // at hundreds of times max_speed -- the oracle does the same -- and a bound of 1.01 max_speed dt then misses a pair: one
        // in 2.8e7 agent-steps of the soak with auto-reset, profiles/r04_soak_parity.txt),
        // so an agent that overlaps this one now was within 2R + 2m of it when the neighbour list was built: if the
        // list is not full it holds every agent within neighbor_dist (>= 2R + 2m required), and if it is full and
        // its farthest member is still beyond 2R + 4m now, its K-th distance then was beyond 2R + 2m -- either way
        // every candidate is in the list and K distances replace the scan of the arena (worth it above 64 agents:
        // C5 80 -> 73 us).  A wave in which some lane cannot conclude that (a clump of more than K agents) scans the
        // arena for those lanes.
        int pairs = 0;
        const float crSq = sqr(R + R);
        float m2 = 0.0f;   // 2 m, a hair wide for the rounding of the update (NaN / infinite speeds fail every test below: full scan)
        if constexpr (LISTP) m2 = 2.0002f * __builtin_sqrtf(__uint_as_float(s_vmax2)) * p.time_step;
        bool scan_all = active && !(LISTP && p.neighbor_dist >= R + R + m2);
        if constexpr (LISTP) if (active && !scan_all) {
            float far2 = 0.0f;
            const int ncnt = (int)(p.counts[q] & 0xFFu);  // read again (this lane wrote it): not kept in a register across the solve
            int jn[KMAX];  // all list entries in flight at once
            static_for<KMAX>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                jn[k] = (k < ncnt) ? ld_idx_t<CA_NBW16(BS)>(p.nb_idx, ((size_t)a * K + k) * N + i) : 0;
            });
            static_for<KMAX>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                if (k < ncnt) {
                    const int j = jn[k];
                    const float d2 = absSq(pos - mk(s_px[lbase + j], s_py[lbase + j]));
                    far2 = d2 > far2 ? d2 : far2;
                    if (j > i && d2 < crSq) ++pairs;
                }
            });
            scan_all = (ncnt == K) && !(far2 > sqr(R + R + 2.0f * m2));
        }
        if constexpr (LISTP) {
            if (__ballot(scan_all) != 0ull && scan_all) {
                pairs = 0;
                for (int j = i + 1; j < N; ++j)
                    if (absSq(pos - mk(s_px[lbase + j], s_py[lbase + j])) < crSq) ++pairs;
            }
        } else {
            // Every unordered pair once, the same work for every lane: agent i tests the agents (i + d) mod N for d = 1 .. (N - 1) / 2,
            // and, N even, the lower half also its antipode d = N / 2 -- N / 2 trips for every lane where the triangular loop
            // `j = i + 1 .. N - 1` cost the WAVE N - 1 trips (lane 0's) for half the useful work.  (x - y)^2 = (y - x)^2 bit for bit, and the
            // per-arena sum is an integer: the same totals.  Four candidates per trip through immediate offsets (the staging above).
            const float2* xs = s_xy2 + (active ? 2 * lbase + i : 0);
            const int H = (N - 1) >> 1;
            auto near = [&](int d) __attribute__((always_inline)) { const float2 o = xs[d]; return absSq(pos - mk(o.x, o.y)) < crSq ? 1 : 0; };
            int d = 1;
            for (; d + 3 <= H; d += 4) pairs += near(d) + near(d + 1) + near(d + 2) + near(d + 3);
            for (; d <= H; ++d) pairs += near(d);
            if (!(N & 1) && N > 1) { const int h = N >> 1; if (i < h) pairs += near(h); }
        }
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
            if (pairs) atomicAdd(&red[1], pairs);
            if (wall) atomicAdd(&red[2], 1);
        }
    }

    CA_STAMP(8);
    CA_PRIO_POINT(7);
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
            c.reward[q] = rew;
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
                c.regoal_count[q] = rc + 1;
            } else {
                done = 1;
                c.arrive_step[q] = steps;
                gx = c.goal2_x[q]; gy = c.goal2_y[q];
                c.agent_done[q] = 1;
            }
            c.goal_x[q] = gx; c.goal_y[q] = gy;
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
        c.agent_done[q] = 0;
        double dx, dy;
        pref_dir64(pos.x, pos.y, gx, gy, &dx, &dy);
        pref = mk((float)dx, (float)dy);
    }
    // sum of rewards: fixed-shape tree inside the wave, then per-arena in lane order
    if ((p.actions || ALAN) && (p.flags & 2u)) {
        double r = active ? (double)(ALAN ? rew_alan : rew) : 0.0;
        const int w = P < 64 ? P : 64;
        // (the guard keeps the tree inside the arena where arenas are packed back to back -- N lanes each, N no power of two;
        // a no-op where an arena has P lanes and the lanes beyond N hold zeros)
        for (int off = w >> 1; off > 0; off >>= 1) { const double t = __shfl_down(r, off, 64); if (!p.dense || i + off < N) r += t; }
        if (active && (i & 63) == 0)
            atomicAdd(reinterpret_cast<double*>(&c.arena_stats[(size_t)a * ST_STRIDE + ST_SUMREW]), r);
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
#if defined(CA_NT_STATE)   // diagnostic (tools/diag/timeline.py): the state leaves through non-temporal stores
        __builtin_nontemporal_store(ox, &c.orient_x[q]); __builtin_nontemporal_store(oy, &c.orient_y[q]);
        __builtin_nontemporal_store(pos.x, &c.pos_x[q]); __builtin_nontemporal_store(pos.y, &c.pos_y[q]);
        __builtin_nontemporal_store(vel.x, &c.vel_x[q]); __builtin_nontemporal_store(vel.y, &c.vel_y[q]);
        if constexpr (!ALAN) { __builtin_nontemporal_store(pref.x, &c.pref_x[q]); __builtin_nontemporal_store(pref.y, &c.pref_y[q]); }
#else
        c.orient_x[q] = ox; c.orient_y[q] = oy;
        c.pos_x[q] = pos.x; c.pos_y[q] = pos.y;
        c.vel_x[q] = vel.x; c.vel_y[q] = vel.y;
        if constexpr (!ALAN) { c.pref_x[q] = pref.x; c.pref_y[q] = pref.y; }   // (ALAN: written above)
#endif
        if (i == 0) {
            unsigned long long* st = c.arena_stats + (size_t)a * ST_STRIDE;
            if (red[1]) st[ST_COLL] += (unsigned)red[1];
            if (red[2]) st[ST_OBST_COLL] += (unsigned)red[2];
            if (red[3]) st[ST_GOALS] += (unsigned)red[3];
            if (all_done) {  // + what a caller that auto-resets wants to know about the episode that ended
                st[ST_EPISODES] += 1;
                st[ST_LASTEP] = ((unsigned long long)(unsigned)steps << 32) | (unsigned)(N - red[0]);
            }
            c.arena_done[a] = all_done ? 1 : 0;
            c.step_count[a] = do_reset ? 0 : steps;
            atomicAdd(&c.arena_steps[a], 1ull);   // (no return value: nothing waits for it at the end of the kernel)
            if (do_reset) c.episode[a] = epi + 1;
        }
    }
    }
    CA_STAMP(11);
}

// ============================================================================================
// reset() for every arena (env.py:461-488): new positions only; velocities, targets and the
// neighbour lists of the last step stay.
// ============================================================================================
__global__ void reset_kernel(const StepArgs p) {
    const ColdK& c = *(ColdK*)p.cold;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p.A * p.N) return;
    const int a = q / p.N, i = q - a * p.N;
    if (p.reset_mask && p.reset_mask[a] == 0) return;
    V2 pos;
    if (p.reset_px) {
        pos = mk(p.reset_px[q], p.reset_py[q]);
    } else {
        double u0, u1;
        rng2(c.seed, c.arena_offset + a, i, RNG_RESET, (uint32_t)c.episode[a], &u0, &u1);
        pos = mk((float)uniform64((double)c.spawn_x0, (double)c.spawn_x1, u0),
                 (float)uniform64((double)c.spawn_y0, (double)c.spawn_y1, u1));
    }
    double dx, dy;
    pref_dir64(pos.x, pos.y, c.goal_x[q], c.goal_y[q], &dx, &dy);
    c.pos_x[q] = pos.x; c.pos_y[q] = pos.y;
    c.pref_x[q] = (float)dx; c.pref_y[q] = (float)dy;
    c.orient_x[q] = (float)dx; c.orient_y[q] = (float)dy;
    c.agent_done[q] = 0;
}
// orientation from scratch (after the caller overwrote positions or goals through ca_set)
__global__ void orient_kernel(const StepArgs p) {
    const ColdK& c = *(ColdK*)p.cold;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p.A * p.N) return;
    double dx, dy;
    pref_dir64(c.pos_x[q], c.pos_y[q], c.goal_x[q], c.goal_y[q], &dx, &dy);
    c.orient_x[q] = (float)dx; c.orient_y[q] = (float)dy;
}
__global__ void reset_arena_kernel(const StepArgs p) {  // after reset_kernel: per-arena counters
    const ColdK& c = *(ColdK*)p.cold;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.A) return;
    if (p.reset_mask && p.reset_mask[a] == 0) return;
    c.step_count[a] = 0;
    c.arena_done[a] = 0;
    c.episode[a] += 1;
}

}  // namespace ca
