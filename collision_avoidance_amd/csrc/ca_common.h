// ca_common.h -- shared types, launch arguments and small device helpers
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
#pragma once
#include "ca_math.h"
#include <utility>

namespace ca {

constexpr int SMAX = 16;      // CA_MAX_OBST_NEIGHBORS (LDS line table; the register-line and quad kernels hold 4)
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

// Arguments of the step / reset kernels.  Kernel arguments are scalar loads that the compiler hoists to the kernel's
// entry and then keeps in SGPRs: with every pointer and constant of the step in one by-value struct the solve kernel
// held ~95 argument registers, spilled ~80 of them to VGPR lanes and paid ~590 v_readlane per wave (7 % of its vector
// instructions).  So the by-value part (StepArgs) carries only what the prologue and the solve read plus what changes
// from call to call; everything the EPILOGUE needs -- the pointers it stores through, the done / reset / re-goal
// constants -- sits in a per-handle block in device memory (StepCold, written once by ca_create) and is loaded, as
// scalar loads, where the epilogue begins.
struct StepCold {
    float *pos_x, *pos_y, *vel_x, *vel_y, *pref_x, *pref_y;
    double *goal_x, *goal_y;        // targets stay fp64 like the reference's Python floats
    const double *goal2_x, *goal2_y;
    float* reward;
    float *orient_x, *orient_y;  // unit vector pos -> goal of the CURRENT state (frame of the observation)
    int *agent_done, *arrive_step, *regoal_count;
    int *step_count, *arena_done, *episode;
    unsigned long long* arena_stats;  // [A][8]
    unsigned long long* arena_steps;  // [A] steps this arena was ADVANCED by a solve kernel (counted in the kernel by the
                                      // arena's owner lane: the evidence of work behind ca_stats.agent_steps)
    unsigned long long* ovf_word;     // page-locked HOST word the device writes: the last obstacle-neighbour list that overflowed
                                      // (note_overflow below); the host turns it into the handle's sticky CA_ERANGE status
    double reward_scale;
    uint64_t seed;
    int64_t arena_offset;
    int max_step, done_mode;
    float done_x_thresh;
    float spawn_x0, spawn_x1, spawn_y0, spawn_y1, goal_x0, goal_x1, goal_y0, goal_y1;
};

// ALAN online learning inside the four-lanes kernel (ca_quad.h): the bandit's arguments, in device memory like StepCold
// (written by ca_alan_configure)
struct AlanCold {
    double *w, *t;          // [A][nA][N] action weights / time since the action's weight was set
    int* action;            // [A*N] the action of the last step
    float* reward;          // [A*N]
    double act_c[32], act_s[32];  // (cos, sin) of every action's angle
    double temp, window, dt, reward_scale;
    int nA;
};

struct StepArgs {
    const float *pos_x, *pos_y, *vel_x, *vel_y, *pref_x, *pref_y;
    const double *goal_x, *goal_y;
    // neighbour lists of the last step, packed: counts [A,N] u16 = agent neighbours | obstacle neighbours << 8;
    // agent-neighbour ids [A,K,N] as u8, or u16 when an arena has more than 256 agents (workgroup > 256 lanes);
    // obstacle-edge ids [A,S,N] as u16
    unsigned short* counts;
    void* nb_idx;
    unsigned short* obst_idx;
    const int* arena_done;
    unsigned long long* arena_stats;  // [A][8]
    const StepCold* cold;  // device memory: the epilogue's pointers and constants (see above)
    const ObstDev* obst;   // the processed edge table(s)
    const int* tab_off;    // null: one table of n_obst edges for every arena; else [A + 1] offsets: arena a owns
                           // edges [tab_off[a], tab_off[a + 1]) and its obstacle-neighbour ids count from tab_off[a]
#ifdef CA_STAMPS
    const int* order;      // diagnostic build only: null, or [blocks]: workgroup b works on the arenas of block order[b]
#endif
    const AlanCold* alan;  // four-lanes kernel, ALAN instantiation: the bandit around every step (ALAN:569-628); else unused
    const double* alan_u;  // ... caller-supplied uniforms [A*N] of a single step, or null: the counter-based RNG
    const float* actions;  // null: orca_step
    const float* reset_px; // explicit reset positions (reset kernel only)
    const float* reset_py;
    const int* reset_mask; // [A] reset only the arenas with a non-zero entry (reset kernels only; null = all)
    unsigned long long* dbg;  // CA_STAMPS diagnostic build only: [waves][16] phase cycle counts
    int n_obst, A, N, P, logP, K, S;
    int a0, a1;  // this launch covers arenas [a0, a1) (chunked launches on several streams)
    // lane -> (arena slot of the workgroup, agent): arenas take LS lanes each -- P (the power of two >= N), or, `dense`, N
    // lanes back to back for arenas within one wave whose N is no power of two (the reference env's own 10-agent arenas: six
    // per wave instead of four).  One formula for both: linv = ceil(2^16 / LS), (lane * linv) >> 16 = lane / LS for every lane
    // of a workgroup (checked by ca_create).
    int LS, apb, linv, dense;
    int nb_hint;   // 1: the agent-neighbour lists in memory are those a solve kernel left (K distinct agents each): the pair
                   // kernel bounds its scan with them (ca_pair.h); 0 after the caller wrote the lists through ca_set
    int T;       // quad kernel, ORCA-only mode: steps advanced by this launch (ca_quad.h); 1 otherwise
    uint32_t flags;
    float time_step, neighbor_dist, time_horizon, time_horizon_obst, radius, max_speed;
};

// An obstacle-neighbour list met more edges in range than it holds: RVO2 keeps them all (env.py:249, 301-318), here the
// farthest are dropped.  Counted per arena (ST_OVERFLOW) by the callers; this makes it LOUD: (global arena, agent, edges in
// range) go to a host-visible word that the next call on the handle -- or its next synchronisation -- reports as CA_ERANGE
// unless the caller opted in (ca_allow_obstacle_overflow).  Rare by construction: a scalar load and one store.
__device__ __forceinline__ void note_overflow(const StepCold* cold, int a, int i, int oin) {
    const unsigned long long g = (unsigned long long)(cold->arena_offset + (int64_t)a) & 0xFFFFFFFFFFull;
    const unsigned long long v = (1ull << 63) | (g << 20) | ((unsigned long long)(i & 0x7FF) << 8) | (unsigned long long)(oin > 255 ? 255 : oin);
    __hip_atomic_store(cold->ovf_word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

enum { ST_EPISODES = 0, ST_COLL = 1, ST_OBST_COLL = 2, ST_GOALS = 3, ST_OVERFLOW = 4, ST_SUMREW = 5, ST_FROZEN = 6, ST_LASTEP = 7, ST_STRIDE = 8 };

// the block of arenas a workgroup works on: its own index (the CA_STAMPS diagnostic build can install a block order
// for the placement experiment of tools/diag/placement.py; the product library has no such indirection)
#ifdef CA_STAMPS
__device__ __forceinline__ int work_block(const StepArgs& p) { return p.order ? p.order[blockIdx.x] : (int)blockIdx.x; }
#else
__device__ __forceinline__ int work_block(const StepArgs&) { return (int)blockIdx.x; }
#endif

__device__ __forceinline__ void lane_slot(const StepArgs& p, int tid, int& la, int& i) {
    la = (int)(__umul24((unsigned)tid, (unsigned)p.linv) >> 16);
    i = tid - (int)__umul24((unsigned)la, (unsigned)p.LS);
}

// CA_F_FREEZE: arenas whose arena_done flag is set are left exactly as they are
__device__ __forceinline__ bool arena_frozen(const StepArgs& p, int a) {
    return (p.flags & 16u) != 0 && a < p.a1 && p.arena_done[a] != 0;
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

// Wave-wide minimum / maximum / inclusive prefix sum through DPP row shifts and row broadcasts (six VALU
// instructions each; EVERY lane of the wave must be active: call them outside divergent code and give idle lanes the
// operation's identity).  The reductions leave their result in lane 63.  (An LDS atomic on ONE address with a value per
// lane is what these replace: the compiler turns that into a loop over the active lanes, nine instructions per lane.)
#define CA_ROW_DPP(old, v, ctrl, rows) __builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rows, 0xF, false)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)CA_ROW_DPP(~0u, v, 0x111, 0xF));  // row_shr:1
    v = min(v, (unsigned)CA_ROW_DPP(~0u, v, 0x112, 0xF));  // row_shr:2
    v = min(v, (unsigned)CA_ROW_DPP(~0u, v, 0x114, 0xF));  // row_shr:4
    v = min(v, (unsigned)CA_ROW_DPP(~0u, v, 0x118, 0xF));  // row_shr:8: lane 15 of a row holds the row's minimum
    v = min(v, (unsigned)CA_ROW_DPP(~0u, v, 0x142, 0xA));  // row_bcast:15 into rows 1 and 3
    v = min(v, (unsigned)CA_ROW_DPP(~0u, v, 0x143, 0xC));  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)CA_ROW_DPP(0u, v, 0x111, 0xF));
    v = max(v, (unsigned)CA_ROW_DPP(0u, v, 0x112, 0xF));
    v = max(v, (unsigned)CA_ROW_DPP(0u, v, 0x114, 0xF));
    v = max(v, (unsigned)CA_ROW_DPP(0u, v, 0x118, 0xF));
    v = max(v, (unsigned)CA_ROW_DPP(0u, v, 0x142, 0xA));
    v = max(v, (unsigned)CA_ROW_DPP(0u, v, 0x143, 0xC));
    return v;
}
__device__ __forceinline__ int wave_prefix_sum(int v) {  // inclusive, every lane
    v += CA_ROW_DPP(0, v, 0x111, 0xF);
    v += CA_ROW_DPP(0, v, 0x112, 0xF);
    v += CA_ROW_DPP(0, v, 0x114, 0xF);
    v += CA_ROW_DPP(0, v, 0x118, 0xF);
    v += CA_ROW_DPP(0, v, 0x142, 0xA);
    v += CA_ROW_DPP(0, v, 0x143, 0xC);
    return v;
}

// packed list entries: the width of an agent-neighbour id is a compile-time property of the kernel (ids of at most
// 256 agents fit a byte), obstacle-edge ids are always 16 bits.  (No run-time width test at the loads: a branch per
// entry keeps the compiler from having a list's loads in flight together -- it cost the observation kernel 11 % --
// and an unaligned 16-bit load with a mask instead of the branch faulted on gfx950.)
#ifdef CA_VAR_NB16   // diagnostic build: 16-bit agent-neighbour ids whatever the arena size
#define CA_NBW16(BS) true
#else
#define CA_NBW16(BS) ((BS) > 256)
#endif
// (the lists live in global memory; saying so keeps the accesses global_* instructions where the compiler has lost
// track of a pointer's origin and would emit flat_* ones)
#define CA_GLOBAL __attribute__((address_space(1)))
template <bool W16>
__device__ __forceinline__ int ld_idx_t(const void* b, size_t k) {
    if constexpr (W16) return (int)((const CA_GLOBAL unsigned short*)b)[k];
    else return (int)((const CA_GLOBAL unsigned char*)b)[k];
}
template <bool W16>
__device__ __forceinline__ void st_idx_t(void* b, size_t k, int v) {
    if constexpr (W16) ((CA_GLOBAL unsigned short*)b)[k] = (unsigned short)v;
    else ((CA_GLOBAL unsigned char*)b)[k] = (unsigned char)v;
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

// Wave priority by progress (step_kernel): the four waves that share a SIMD are served oldest first, so they finish one
// after the other and the last one runs alone -- at a lone wave's issue rate -- for the final fifth of the kernel.  A wave
// that LOWERS its priority at each phase boundary lets the waves behind it catch up, so the four stay within a phase of
// each other and finish together (C3: step_kernel 81.9 -> 71.2 us; profiles/r03_b_wave_priority_experiment.txt).
// Points along step_kernel: 1 after the neighbour search, 2 after the obstacle lines, 3 after the agent lines, 4 in the
// middle of LP2, 5 after LP2, 6 after LP3 + integration, 7 after the collision statistics.  A wave starts at priority 3
// and drops to 2, 1, 0 at the points CA_PRIO_B1 < CA_PRIO_B2 < CA_PRIO_B3 (0 = never: no priority instructions at all).
#ifndef CA_PRIO_B1
#define CA_PRIO_B1 2
#define CA_PRIO_B2 4
#define CA_PRIO_B3 6
#endif
#if CA_PRIO_B1 > 0
#define CA_PRIO_START() __builtin_amdgcn_s_setprio(3)
#define CA_PRIO_POINT(x)                                                    \
    do {                                                                    \
        if ((x) == CA_PRIO_B1) __builtin_amdgcn_s_setprio(2);               \
        if ((x) == CA_PRIO_B2) __builtin_amdgcn_s_setprio(1);               \
        if ((x) == CA_PRIO_B3) __builtin_amdgcn_s_setprio(0);               \
    } while (0)
#else
#define CA_PRIO_START() do { } while (0)
#define CA_PRIO_POINT(x) do { } while (0)
#endif

#ifdef CA_STAMPS  // diagnostic build: per-wave cycle count of each phase (never in the product library)
#if CA_STAMPS >= 2   // wall-clock variant: the 100 MHz device-wide counter (wave timelines across CUs)
#define CA_STAMP_CLOCK() __builtin_amdgcn_s_memrealtime()
#else                // per-CU shader-clock counter (phase shares inside a wave)
#define CA_STAMP_CLOCK() __builtin_amdgcn_s_memtime()
#endif
// CA_STAMPS == 4 (tools/diag/timeline.py) and == 3 (tools/diag/placement.py): only the wave's FIRST stamp (12: the head of the fused neighbour search) and its LAST
// (11) are taken -- a timeline of wave starts and ends against the dispatch, from a kernel that is otherwise the product's
#define CA_STAMP(k)                                                                      \
    do {                                                                                 \
        if (CA_STAMPS < 3 || (k) == 11 || (k) == 12) {                                  \
        __builtin_amdgcn_sched_barrier(0);                                               \
        const unsigned long long _t = CA_STAMP_CLOCK();                                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                              \
        if ((threadIdx.x & 63) == 0 && threadIdx.x < BS && p.dbg)                        \
            p.dbg[((size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6)) * 16 + (k)] = _t; \
        __builtin_amdgcn_sched_barrier(0);                                               \
        }                                                                                \
    } while (0)
#if CA_STAMPS == 3   // placement variant: slots 2 and 3 of a wave's record hold its hardware id (HW_ID, XCC_ID)
#define CA_STAMP_HWID()                                                                                    \
    do {                                                                                                   \
        unsigned _hw, _xcc;                                                                                \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(_hw));                                  \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(_xcc));                                \
        if ((threadIdx.x & 63) == 0 && p.dbg) {                                                            \
            p.dbg[((size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6)) * 16 + 2] = _hw;                   \
            p.dbg[((size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6)) * 16 + 3] = _xcc;                  \
        }                                                                                                  \
    } while (0)
#endif
#else
#define CA_STAMP(k) do { } while (0)
#endif
#ifndef CA_STAMP_HWID
#define CA_STAMP_HWID() do { } while (0)
#endif

}  // namespace ca
