// ca_env.hip -- host side of libcaenv.so: the C ABI declared in include/ca_env.h.
// Owns the device buffers (struct-of-arrays, fp32/int32), the obstacle table and the launch
// geometry; every entry point enqueues work on the handle's stream.  No CPU fallback exists.
#include "../../include/ca_env.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ca_kernels.h"

using namespace ca;

struct ca_env {
    ca_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // [A*N] fp32
    float *pos_x = nullptr, *pos_y = nullptr, *vel_x = nullptr, *vel_y = nullptr, *pref_x = nullptr,
          *pref_y = nullptr, *reward = nullptr, *orient_x = nullptr, *orient_y = nullptr;
    bool orient_valid = false;  // orient_x/y match pos/goal (false after the caller edits them)
    double *goal_x = nullptr, *goal_y = nullptr, *goal2_x = nullptr, *goal2_y = nullptr;  // fp64 targets
    int *agent_done = nullptr, *arrive_step = nullptr, *regoal_count = nullptr;
    // neighbour lists, packed (ca_common.h StepArgs): counts u16 [A,N]; indices u8 or u16 [A,K,N] / [A,S,N]
    unsigned short* counts = nullptr;
    void* nb_idx = nullptr;
    unsigned short* obst_idx = nullptr;
    int nidx16 = 0;
    int* cvt_buf = nullptr;  // staging of the i32 image the ABI shows for the packed fields (ca_get / ca_set)
    size_t cvt_cap = 0;
    int *step_count = nullptr, *arena_done = nullptr, *episode = nullptr;
    unsigned long long* arena_stats = nullptr;
    unsigned long long* arena_steps = nullptr;  // [A] steps each arena was advanced, counted by the solve kernels
    float* obs = nullptr;
    bool obs_external = false;
    // what a step hands back -- observation | reward | arena_done | step_count -- lies in ONE device allocation in this order, so
    // that the host-array form of a step (ca_step_packed) fetches all of it with one copy (obs points elsewhere after ca_bind_obs)
    float* slab = nullptr;
    unsigned long long* dbg = nullptr;  // CA_STAMPS diagnostic build only
    unsigned long long* dbg_obs = nullptr;
    float *tmp_x = nullptr, *tmp_y = nullptr;  // staging for explicit reset positions / host actions
    // ALAN online learning (ca_alan_configure): [A*N][n_actions] fp64 weights / times, [A*N] action,
    // [A*N][4] fp64 directions of the step in flight
    double *alan_w = nullptr, *alan_t = nullptr, *alan_dirs = nullptr, *alan_u = nullptr;
    int* alan_action = nullptr;
    AlanCold* d_alan = nullptr;
    bool alan_fused = false;      // ca_alan_step / ca_alan_rollout run as ONE launch of the four-lanes kernel
    bool alan_lane = false;       // ca_alan_step runs as ONE launch of the register-line lane kernel (its ALAN instantiation)   // the bandit's arguments for the four-lanes kernel (ca_common.h)
    int* mask_buf = nullptr;  // staging for ca_reset_masked's host mask
    std::vector<std::pair<void*, size_t>> host_allocs;   // page-locked host buffers handed out by ca_host_alloc (freed by ca_host_free / ca_destroy)
    int n_actions = 0;
    double act_c[CA_ALAN_MAX_ACTIONS], act_s[CA_ALAN_MAX_ACTIONS];
    double alan_temp = 0.2, alan_window = 2.0, alan_dt = 1.0 / 60.0;
    ObstDev* d_obst = nullptr;
    std::vector<ObstDev> h_obst;     // every table, concatenated
    std::vector<int> h_tab_off;      // empty: one table for all arenas; else [A + 1] offsets into h_obst
    int* d_tab_off = nullptr;
    StepCold* d_cold = nullptr;      // the epilogue's arguments (ca_common.h)
    // obstacle-neighbour overflow made loud: a page-locked host word the kernels write (ca_common.h note_overflow); unless the
    // caller opted in (ca_allow_obstacle_overflow) it is the handle's sticky CA_ERANGE status (overflow_status below)
    unsigned long long* ovf_host = nullptr;
    bool allow_overflow = false;
    int* d_order = nullptr;          // [grid] block order of the solve kernel (null: identity)
    int P = 1, logP = 0, BS = 64, grid = 1, K = 0, S = 1;
    bool obs_dense_on = false;
    bool lists_trusted = true;   // the neighbour lists in memory were written by the kernels (not by the caller through ca_set)
    int LS = 1, apb = 1, linv = 65536, dense = 0;   // lanes per arena, arenas per workgroup, ceil(2^16 / LS), packed back to back (ca_common.h StepArgs)
    // diagnostic environment switches, read ONCE by ca_create (a handle never changes behaviour because the environment did):
    // -1 = unset, else the first character's digit
    struct { int reg_lines = -1, nbr_help = -1, pair = -1, pair_min = 129, alan_fused = -1; } sw;
    int BSn = 64, grid_n = 1;  // the neighbour kernel's own workgroup size
    bool fuse_nbr = true;       // neighbour search at the head of the solve kernel (default) or as its own launch
    int ST = 0, KT = 16;  // solve-kernel variant: ST > 0 = register lines with ST obstacle slots; KT = KMAX
    int SMX = 4;          // ... and the capacity of its obstacle-neighbour list (4, or 16: agents with more than ST are solved apart)
    int max_edges = 0;    // edges of the largest installed obstacle table (the variant depends on it: pick_variant)
    int n_cus = 256;      // compute units of the device (pick_variant: is the batch resident with the LDS line table?)
    bool help = false;     // large arenas: helper lanes in the uniform-grid neighbour scan (ca_nbr.h, HELP = 2)
    bool pair = false;     // large arenas: two lanes per agent for the whole step (ca_pair.h); replaces `help` where chosen
    size_t lds_p = 0;
    bool quad = false;     // four lanes per agent (ca_quad.h): small batches / small arenas
    bool quad_roll = false;  // ... for ca_rollout's one-launch-for-T-steps form (pays a little longer than for single steps)
    int BSq = 64, grid_q = 1, SQ = 4;   // SQ: obstacle-neighbour capacity of the quad variant (4 or 16)
    size_t lds_q = 0;
    size_t lds = 0;
    uint64_t steps_done = 0;  // env steps enqueued (profiling cadence only: ca_stats.agent_steps is counted in the kernels)
    float rays[32], oct[32];
    // opt-in per-kernel timing (ca_profile): event pairs recorded around launches, drained on read
    bool profiling = false;   // events are recorded for the current step
    int prof_period = 0;      // 0 = off, k = every k-th step
    struct Span { hipEvent_t t0, t1; int kind; int steps; };  // steps: env steps advanced by the timed launch (ca_rollout: T)
    std::vector<Span> spans;
    std::vector<hipEvent_t> free_events;
    std::string err;
};

static thread_local std::string g_create_err;

static int fail(ca_env* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    else g_create_err = buf;
    return code;
}
#define HIPCHK(e, call)                                                                         \
    do {                                                                                        \
        hipError_t _r = (call);                                                                 \
        if (_r != hipSuccess) return fail(e, CA_EHIP, "%s failed: %s", #call, hipGetErrorString(_r)); \
    } while (0)

// RVO2 keeps EVERY obstacle edge in range of an agent (env.py:249, 301-318 iterate them all); the lists here hold
// max_obst_neighbors (<= CA_MAX_OBST_NEIGHBORS) and drop the farthest beyond that -- a deviation that must not pass silently.
// The kernels report the last overflow through a host-visible word; every call that advances the environment returns
// CA_ERANGE from then on (like a sticky device error: the call AFTER the kernel that overflowed has finished sees it, and the
// calls that synchronise -- ca_sync, ca_step_packed -- see it at once), until ca_reset_stats clears it or the caller accepts
// the truncation with ca_allow_obstacle_overflow(env, 1).  ca_get / ca_get_stats keep working (obst_overflow counts).
static int overflow_status(ca_env* e, const char* where) {
    if (e->allow_overflow || !e->ovf_host) return CA_OK;
    const unsigned long long v = *(volatile unsigned long long*)e->ovf_host;
    if (!(v >> 63)) return CA_OK;
    return fail(e, CA_ERANGE, "%s: an obstacle-neighbour list overflowed: arena %lld, agent %d had %d%s obstacle edges in range but "
                "max_obst_neighbors=%d, so the farthest were dropped (RVO2 keeps every edge in range: collision_avoidence_env.py:249, "
                "301-318).  Raise max_obst_neighbors (at most %d), coarsen the world's polylines, or accept the truncation with "
                "ca_allow_obstacle_overflow(env, 1); ca_reset_stats clears this status", where,
                (long long)((v >> 20) & 0xFFFFFFFFFFull), (int)((v >> 8) & 0x7FF), (int)(v & 0xFF), (v & 0xFF) == 255 ? "+" : "",
                e->S, CA_MAX_OBST_NEIGHBORS);
}

enum { KIND_NBR = 0, KIND_STEP = 1, KIND_OBS = 2, KIND_RESET = 3 };
enum { CA_ROLLOUT_MAX_T = 256 };  // steps per launch of the one-launch rollout (ca_rollout)  // KIND_RESET also times the small ALAN kernels
static hipEvent_t prof_event(ca_env* e) {
    if (!e->free_events.empty()) { hipEvent_t ev = e->free_events.back(); e->free_events.pop_back(); return ev; }
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}
// Kernel times: a sampled launch carries a start and a stop event ON ITS OWN DISPATCH (hipExtLaunchKernel: the
// timestamps of the dispatch packet's completion signal), not events recorded around it in the stream -- a recorded event
// is a barrier packet of its own and costs the step loop ~2.5 us each, 4 % of a C3 step when every second step is sampled.
struct ProfScope {
    ca_env* e; hipEvent_t t0 = nullptr, t1 = nullptr; int kind; int steps;
    ProfScope(ca_env* env, int k, int n_steps = 1) : e(env), kind(k), steps(n_steps) {
        if (!e->profiling) return;
        t0 = prof_event(e);
        t1 = t0 ? prof_event(e) : nullptr;
        if (t0 && !t1) { e->free_events.push_back(t0); t0 = nullptr; }
    }
    ~ProfScope() {
        if (t0) e->spans.push_back({t0, t1, kind, steps});
    }
};
template <class... Args>
static void launch_k(const ProfScope& ps, void (*kernel)(Args...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                     Args... args) {
    if (ps.t0) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, ps.t0, ps.t1, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}

static size_t AN(const ca_env* e) { return (size_t)e->cfg.n_arenas * e->cfg.n_agents; }

struct FieldInfo {
    void* ptr;
    size_t bytes;
    bool writable;
};
static FieldInfo field_info(ca_env* e, int f) {
    const size_t an = AN(e), A = e->cfg.n_arenas;
    switch (f) {
        case CA_FLD_POS_X: return {e->pos_x, an * 4, true};
        case CA_FLD_POS_Y: return {e->pos_y, an * 4, true};
        case CA_FLD_VEL_X: return {e->vel_x, an * 4, true};
        case CA_FLD_VEL_Y: return {e->vel_y, an * 4, true};
        case CA_FLD_PREF_X: return {e->pref_x, an * 4, true};
        case CA_FLD_PREF_Y: return {e->pref_y, an * 4, true};
        case CA_FLD_GOAL_X: return {e->goal_x, an * 8, true};
        case CA_FLD_GOAL_Y: return {e->goal_y, an * 8, true};
        case CA_FLD_GOAL2_X: return {e->goal2_x, an * 8, true};
        case CA_FLD_GOAL2_Y: return {e->goal2_y, an * 8, true};
        case CA_FLD_REWARD: return {e->reward, an * 4, true};   // (writable for set_state: a frozen arena keeps its last reward)
        case CA_FLD_AGENT_DONE: return {e->agent_done, an * 4, true};
        case CA_FLD_ARRIVE_STEP: return {e->arrive_step, an * 4, true};
        // packed in device memory; the ABI shows them as i32 (ca_get / ca_set convert, ca_field_ptr refuses)
        case CA_FLD_NB_COUNT: return {e->counts, an * 4, true};
        case CA_FLD_NB_IDX: return {e->nb_idx, an * 4 * (size_t)(e->K > 0 ? e->K : 1), true};
        case CA_FLD_OBST_COUNT: return {e->counts, an * 4, true};
        case CA_FLD_OBST_IDX: return {e->obst_idx, an * 4 * (size_t)e->S, true};
        case CA_FLD_OBS: return {e->obs, an * CA_OBS_DIM * 4, false};
        case CA_FLD_STEP_COUNT: return {e->step_count, A * 4, true};
        case CA_FLD_ARENA_DONE: return {e->arena_done, A * 4, true};
        case CA_FLD_EPISODE: return {e->episode, A * 4, true};
        case CA_FLD_REGOAL_COUNT: return {e->regoal_count, an * 4, true};
        case CA_FLD_ALAN_WEIGHTS: return {e->alan_w, an * 8 * (size_t)e->n_actions, true};
        case CA_FLD_ALAN_TIMES: return {e->alan_t, an * 8 * (size_t)e->n_actions, true};
        case CA_FLD_ALAN_ACTION: return {e->alan_action, an * 4, true};   // (writable for set_state: a frozen arena keeps its last action)
        case CA_FLD_ARENA_STATS: return {e->arena_stats, A * ST_STRIDE * 8, false};
        default: return {nullptr, 0, false};
    }
}

static void host_tables(ca_env* e) {
    // env.py:321-332 ray end points; env.py:335-350 octagon chords (fp64 like the reference's
    // Python floats, then rounded once to fp32)
    const double nd = (double)e->cfg.neighbor_dist, r = (double)e->cfg.radius;
    for (int i = 0; i < 16; ++i) {
        const double th = i * (2.0 * M_PI / 16);
        e->rays[2 * i] = (float)(nd * std::cos(th));
        e->rays[2 * i + 1] = (float)(-nd * std::sin(th));
    }
    double first[2] = {r * std::cos(0.0), -r * std::sin(0.0)}, cur[2] = {first[0], first[1]};
    int k = 0;
    for (int i = 1; i < 8; ++i, ++k) {
        const double th = i * (2.0 * M_PI / 8);
        const double nx = r * std::cos(th), ny = -r * std::sin(th);
        e->oct[4 * k] = (float)cur[0]; e->oct[4 * k + 1] = (float)cur[1];
        e->oct[4 * k + 2] = (float)nx; e->oct[4 * k + 3] = (float)ny;
        cur[0] = nx; cur[1] = ny;
    }
    e->oct[4 * k] = (float)cur[0]; e->oct[4 * k + 1] = (float)cur[1];
    e->oct[4 * k + 2] = (float)first[0]; e->oct[4 * k + 3] = (float)first[1];
}

// the per-handle block of epilogue arguments (ca_common.h StepCold); written once, at the end of ca_create
static void fill_cold(const ca_env* e, StepCold& c) {
    const ca_config& g = e->cfg;
    c.pos_x = e->pos_x; c.pos_y = e->pos_y; c.vel_x = e->vel_x; c.vel_y = e->vel_y;
    c.pref_x = e->pref_x; c.pref_y = e->pref_y; c.goal_x = e->goal_x; c.goal_y = e->goal_y;
    c.goal2_x = e->goal2_x; c.goal2_y = e->goal2_y; c.reward = e->reward;
    c.orient_x = e->orient_x; c.orient_y = e->orient_y;
    c.agent_done = e->agent_done; c.arrive_step = e->arrive_step; c.regoal_count = e->regoal_count;
    c.step_count = e->step_count; c.arena_done = e->arena_done; c.episode = e->episode;
    c.arena_stats = e->arena_stats; c.arena_steps = e->arena_steps; c.ovf_word = e->ovf_host;
    c.reward_scale = g.reward_scale; c.seed = g.seed; c.arena_offset = g.arena_offset;
    c.max_step = g.max_step; c.done_mode = g.done_mode; c.done_x_thresh = g.done_x_thresh;
    c.spawn_x0 = g.spawn_x0; c.spawn_x1 = g.spawn_x1; c.spawn_y0 = g.spawn_y0; c.spawn_y1 = g.spawn_y1;
    c.goal_x0 = g.goal_x0; c.goal_x1 = g.goal_x1; c.goal_y0 = g.goal_y0; c.goal_y1 = g.goal_y1;
}

static void fill_args(ca_env* e, StepArgs& a, const float* actions, uint32_t flags) {
    const ca_config& c = e->cfg;
    a.pos_x = e->pos_x; a.pos_y = e->pos_y; a.vel_x = e->vel_x; a.vel_y = e->vel_y;
    a.pref_x = e->pref_x; a.pref_y = e->pref_y; a.goal_x = e->goal_x; a.goal_y = e->goal_y;
    a.counts = e->counts; a.nb_idx = e->nb_idx; a.obst_idx = e->obst_idx;
    a.arena_done = e->arena_done; a.arena_stats = e->arena_stats; a.cold = e->d_cold;
    a.obst = e->d_obst; a.tab_off = e->d_tab_off; a.actions = actions; a.alan = nullptr; a.alan_u = nullptr;
#ifdef CA_STAMPS
    a.order = e->fuse_nbr ? e->d_order : nullptr;  // (the order is sized for the solve kernel's grid)
#endif
    a.reset_px = nullptr; a.reset_py = nullptr; a.reset_mask = nullptr; a.dbg = e->dbg;
    a.n_obst = e->h_tab_off.empty() ? (int)e->h_obst.size() : 0; a.A = c.n_arenas; a.N = c.n_agents; a.P = e->P; a.logP = e->logP;
    a.K = e->K; a.S = e->S; a.flags = flags; a.a0 = 0; a.a1 = c.n_arenas; a.T = 1;
    a.LS = e->LS; a.apb = e->apb; a.linv = e->linv; a.dense = e->dense; a.nb_hint = e->lists_trusted ? 1 : 0;
    a.time_step = c.time_step; a.neighbor_dist = c.neighbor_dist; a.time_horizon = c.time_horizon;
    a.time_horizon_obst = c.time_horizon_obst; a.radius = c.radius; a.max_speed = c.max_speed;
}

template <int KMAX, int SM>
static void launch_nbr_k(ca_env* e, const StepArgs& a_in) {
    const dim3 grid(e->grid_n), block(e->BSn);
    ProfScope ps(e, KIND_NBR);
    StepArgs a = a_in;
    a.apb = e->BSn / e->P; a.LS = e->P; a.linv = 65536 / e->P; a.dense = 0;   // (the stand-alone neighbour kernel may run with a workgroup size of its own: CA_NBR_BS)
    switch (e->BSn) {
        case 64: launch_k(ps, nbr_kernel<KMAX, 64, SM>, grid, block, 0, e->stream, a); break;
        case 128: launch_k(ps, nbr_kernel<KMAX, 128, SM>, grid, block, 0, e->stream, a); break;
        case 256: launch_k(ps, nbr_kernel<KMAX, 256, SM>, grid, block, 0, e->stream, a); break;
        case 512: launch_k(ps, nbr_kernel<KMAX, 512, SM>, grid, block, 0, e->stream, a); break;
        default: launch_k(ps, nbr_kernel<KMAX, 1024, SM>, grid, block, 0, e->stream, a); break;
    }
}
template <int KMAX, int ST, bool FUSE>
static hipError_t launch_step_kf(ca_env* e, const StepArgs& a) {
    if (!FUSE) launch_nbr_k<KMAX, (ST > 0 ? ST : SMAX)>(e, a);  // neighbour search as a launch of its own (diagnostic: CA_FUSE_NBR=0)
    const dim3 grid(e->grid), block(e->BS);
    ProfScope ps(e, KIND_STEP);
    if constexpr (FUSE && ST > 0) {
        if (e->pair) {  // two lanes per agent (ca_pair.h): one arena per workgroup of 2 P lanes
            if (e->BS == 256) launch_k(ps, pair_kernel<KMAX, 256>, grid, dim3(512), e->lds_p, e->stream, a);
            else launch_k(ps, pair_kernel<KMAX, 512>, grid, dim3(1024), e->lds_p, e->stream, a);
            return hipGetLastError();
        }
        if (e->help) {  // twice the lanes: the upper half helps in the neighbour scan of its arena and ends (ca_nbr.h)
            if (e->BS == 256) launch_k(ps, step_kernel<KMAX, 256, ST, true, 2>, grid, dim3(512), e->lds, e->stream, a);
            else launch_k(ps, step_kernel<KMAX, 512, ST, true, 2>, grid, dim3(1024), e->lds, e->stream, a);
            return hipGetLastError();
        }
    }
    if constexpr (FUSE && KMAX <= 10) {
        if (a.alan != nullptr) {   // the ALAN bandit inside the launch (alan_pick allowed it: alan_lane)
            if constexpr (ST > 0) {
                if (e->SMX > ST) {   // (obstacle-neighbour lists of up to 16: "congested", the doorway world)
                    if (e->BS == 64) launch_k(ps, step_kernel<KMAX, 64, ST, true, 1, 16, true>, grid, block, e->lds, e->stream, a);
                    else launch_k(ps, step_kernel<KMAX, 128, ST, true, 1, 16, true>, grid, block, e->lds, e->stream, a);
                } else {
                    if (e->BS == 64) launch_k(ps, step_kernel<KMAX, 64, ST, true, 1, ST, true>, grid, block, e->lds, e->stream, a);
                    else launch_k(ps, step_kernel<KMAX, 128, ST, true, 1, ST, true>, grid, block, e->lds, e->stream, a);
                }
            } else {                 // (the LDS line table: "deadlock", "blocks")
                if (e->BS == 64) launch_k(ps, step_kernel<KMAX, 64, 0, true, 1, SMAX, true>, grid, block, e->lds, e->stream, a);
                else launch_k(ps, step_kernel<KMAX, 128, 0, true, 1, SMAX, true>, grid, block, e->lds, e->stream, a);
            }
            return hipGetLastError();
        }
    }
    if constexpr (FUSE && ST > 0) {
        if (e->SMX > ST) {  // register lines, obstacle-neighbour lists of up to 16 (agents with more than ST are solved apart)
            switch (e->BS) {
                case 64: launch_k(ps, step_kernel<KMAX, 64, ST, true, 1, 16>, grid, block, e->lds, e->stream, a); break;
                case 128: launch_k(ps, step_kernel<KMAX, 128, ST, true, 1, 16>, grid, block, e->lds, e->stream, a); break;
                case 256: launch_k(ps, step_kernel<KMAX, 256, ST, true, 1, 16>, grid, block, e->lds, e->stream, a); break;
                case 512: launch_k(ps, step_kernel<KMAX, 512, ST, true, 1, 16>, grid, block, e->lds, e->stream, a); break;
                default: launch_k(ps, step_kernel<KMAX, 1024, ST, true, 1, 16>, grid, block, e->lds, e->stream, a); break;
            }
            return hipGetLastError();
        }
    }
    switch (e->BS) {  // (neighbour search +) lines + LP + integration + reward/done
        case 64: launch_k(ps, step_kernel<KMAX, 64, ST, FUSE>, grid, block, e->lds, e->stream, a); break;
        case 128: launch_k(ps, step_kernel<KMAX, 128, ST, FUSE>, grid, block, e->lds, e->stream, a); break;
        case 256: launch_k(ps, step_kernel<KMAX, 256, ST, FUSE>, grid, block, e->lds, e->stream, a); break;
        case 512: launch_k(ps, step_kernel<KMAX, 512, ST, FUSE>, grid, block, e->lds, e->stream, a); break;
        default: launch_k(ps, step_kernel<KMAX, 1024, ST, FUSE>, grid, block, e->lds, e->stream, a); break;
    }
    return hipGetLastError();
}
template <int KMAX, int ST>
static hipError_t launch_step_k(ca_env* e, const StepArgs& a) {
#ifdef CA_WITH_UNFUSED_NBR   // diagnostic build: the neighbour search as a launch of its own (CA_FUSE_NBR=0 / CA_NBR_BS)
    if (!e->fuse_nbr) return launch_step_kf<KMAX, ST, false>(e, a);
#endif
    return launch_step_kf<KMAX, ST, true>(e, a);
}
template <int KMAX, int SQ, bool ALAN>
static const void* quad_fn_k(int BS) {
    switch (BS) {
        case 64: return reinterpret_cast<const void*>(&quad_kernel<KMAX, 64, SQ, ALAN>);
        case 128: return reinterpret_cast<const void*>(&quad_kernel<KMAX, 128, SQ, ALAN>);
        case 256: return reinterpret_cast<const void*>(&quad_kernel<KMAX, 256, SQ, ALAN>);
        default: return reinterpret_cast<const void*>(&quad_kernel<KMAX, 512, SQ, ALAN>);
    }
}
static const void* quad_fn(const ca_env* e, bool alan = false) {
    if (alan) {
        if (e->KT == 5) return e->SQ == 4 ? quad_fn_k<5, 4, true>(e->BSq) : quad_fn_k<5, 16, true>(e->BSq);
        return e->SQ == 4 ? quad_fn_k<10, 4, true>(e->BSq) : quad_fn_k<10, 16, true>(e->BSq);
    }
    if (e->KT == 5) return e->SQ == 4 ? quad_fn_k<5, 4, false>(e->BSq) : quad_fn_k<5, 16, false>(e->BSq);
    return e->SQ == 4 ? quad_fn_k<10, 4, false>(e->BSq) : quad_fn_k<10, 16, false>(e->BSq);
}
// neighbour search + lines + LP + integration + reward/done, four lanes per agent, a.T steps
static hipError_t launch_quad(ca_env* e, const StepArgs& a) {
    ProfScope ps(e, KIND_STEP, a.T > 1 ? a.T : 1);
    StepArgs arg = a;
    void* params[] = {&arg};
    const bool alan = a.alan != nullptr;   // the bandit inside the launch (ca_alan_step / ca_alan_rollout)
    const size_t lds = alan ? quad_lds_bytes(e->BSq, e->KT, e->SQ, e->n_actions) : e->lds_q;
    if (ps.t0) return hipExtLaunchKernel(quad_fn(e, alan), dim3(e->grid_q), dim3(e->BSq), params, lds, e->stream, ps.t0, ps.t1, 0);
    return hipLaunchKernel(quad_fn(e, alan), dim3(e->grid_q), dim3(e->BSq), params, lds, e->stream);
}
static hipError_t launch_step_any(ca_env* e, const StepArgs& a);
static hipError_t launch_step(ca_env* e, const StepArgs& a) {
    const hipError_t r = launch_step_any(e, a);
    if (!(a.flags & CA_F_FREEZE)) e->lists_trusted = true;   // every arena's lists are this launch's now (frozen arenas keep theirs)
    return r;
}
static hipError_t launch_step_any(ca_env* e, const StepArgs& a) {
    if (e->quad || (a.T > 1 && e->quad_roll) || (a.alan != nullptr && !e->alan_lane)) return launch_quad(e, a);
    if (e->ST > 0) return e->KT == 5 ? launch_step_k<5, 4>(e, a) : launch_step_k<10, 4>(e, a);
    if (e->K <= 5) return launch_step_k<5, 0>(e, a);
    if (e->K <= 10) return launch_step_k<10, 0>(e, a);
    return launch_step_k<16, 0>(e, a);
}

template <int KMAX, int BS, int ST>
static hipError_t set_lds_attr(size_t lds) {
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<KMAX, BS, ST, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (r != hipSuccess) return r;
    if constexpr (ST > 0) {
        r = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<KMAX, BS, ST, true, 1, 16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (r != hipSuccess) return r;
    }
    if constexpr (ST == 0 && KMAX <= 10 && BS <= 128) {   // the ALAN instantiation of the LDS line table (two waves: 57 KB)
        r = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<KMAX, BS, 0, true, 1, SMAX, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (r != hipSuccess) return r;
    }
#ifdef CA_WITH_UNFUSED_NBR
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<KMAX, BS, ST, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
    return r;
}
template <int KMAX, int ST>
static hipError_t set_lds_attr_k(int BS, size_t lds) {
    switch (BS) {
        case 64: return set_lds_attr<KMAX, 64, ST>(lds);
        case 128: return set_lds_attr<KMAX, 128, ST>(lds);
        case 256: return set_lds_attr<KMAX, 256, ST>(lds);
        case 512: return set_lds_attr<KMAX, 512, ST>(lds);
        default: return set_lds_attr<KMAX, 1024, ST>(lds);
    }
}

typedef void (*obs_fn_t)(const ObsArgs);
static obs_fn_t obs_fn(int obs_bs, bool w16, bool dense = false) {  // workgroup size x width of the stored agent-neighbour ids
    if (dense) return obs_kernel<256, false, true>;   // (arenas of fewer than 16 agents: 256 lanes, 8-bit ids)
    switch (obs_bs) {
        case 1024: return w16 ? obs_kernel<1024, true> : obs_kernel<1024, false>;
        case 512: return w16 ? obs_kernel<512, true> : obs_kernel<512, false>;
        case 128: return w16 ? obs_kernel<128, true> : obs_kernel<128, false>;
        case 64: return w16 ? obs_kernel<64, true> : obs_kernel<64, false>;
        default: return w16 ? obs_kernel<256, true> : obs_kernel<256, false>;
    }
}

// Arenas of fewer than 16 agents: the 16 agent groups of an observation workgroup take 16 consecutive agents of the batch (the
// reference env's own 10-agent arenas would otherwise leave 6 of 16 groups idle).  CA_OBS_DENSE=0: one arena per workgroup.
static bool obs_dense(const ca_env* e) { return e->obs_dense_on; }   // (latched by ca_create)
static int obs_nstage(const ca_env* e) { return obs_dense(e) ? 16 + 2 * e->cfg.n_agents : e->cfg.n_agents; }

static hipError_t launch_obs(ca_env* e) {
    if (!e->orient_valid) {  // positions or goals were edited from outside: re-derive the frame
        StepArgs a;
        fill_args(e, a, nullptr, 0);
        const unsigned an = (unsigned)AN(e);
        hipLaunchKernelGGL(orient_kernel, dim3((an + 255) / 256), dim3(256), 0, e->stream, a);
        hipError_t r = hipGetLastError();
        if (r != hipSuccess) return r;
        e->orient_valid = true;
    }
    ObsArgs o;
    o.pos_x = e->pos_x; o.pos_y = e->pos_y; o.vel_x = e->vel_x; o.vel_y = e->vel_y;
    o.orient_x = e->orient_x; o.orient_y = e->orient_y; o.counts = e->counts; o.nb_idx = e->nb_idx;
    o.obst_idx = e->obst_idx; o.obst = e->d_obst; o.tab_off = e->d_tab_off;
    o.obs = e->obs;
    o.A = e->cfg.n_arenas; o.N = e->cfg.n_agents; o.S = e->S;
    o.K = e->K > 0 ? e->K : 1;  // nb_idx is allocated with one column when K == 0; counts are all zero
    const int obs_bs = obs_block_threads(o.N), apb = obs_bs / 16;
    o.bpa = (o.N + apb - 1) / apb;
    o.paircap = 16 * (e->K + e->S);
    o.a0 = 0; o.dbg = e->dbg_obs;
    o.dense = obs_dense(e) ? 1 : 0;
    o.nstage_max = obs_nstage(e);
    o.xcd = (o.A % 8 == 0 && !o.dense) ? 1 : 0;  // the arena's observation on the XCD (workgroup index mod 8) whose solve workgroup wrote its state
    o.radius = e->cfg.radius;
    memcpy(o.rays, e->rays, sizeof o.rays);
    memcpy(o.oct, e->oct, sizeof o.oct);
    const dim3 grid(o.dense ? (unsigned)(((size_t)o.A * o.N + apb - 1) / apb) : (unsigned)((size_t)o.A * o.bpa)), block(obs_bs);
    const size_t lds = obs_lds_bytes(o.nstage_max, obs_bs, o.paircap);
    ProfScope ps(e, KIND_OBS);
    launch_k(ps, obs_fn(obs_bs, e->nidx16 != 0, o.dense != 0), grid, block, lds, e->stream, o);
    return hipGetLastError();
}

// reset_kernel + reset_arena_kernel, each timed on its own dispatch (kind 3) when the step is sampled
static hipError_t launch_reset(ca_env* e, const StepArgs& a) {
    const unsigned an = (unsigned)AN(e);
    {
        ProfScope ps(e, KIND_RESET);
        launch_k(ps, reset_kernel, dim3((an + 255) / 256), dim3(256), 0, e->stream, a);
    }
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) return r;
    {
        ProfScope ps(e, KIND_RESET);
        launch_k(ps, reset_arena_kernel, dim3((e->cfg.n_arenas + 255) / 256), dim3(256), 0, e->stream, a);
    }
    return hipGetLastError();
}

// What sim.processObstacles() (env.py:123, ALAN:209) does to the vertex table in the RVO2 library: its
// obstacle BSP tree picks, per node, the edge whose supporting line balances the remaining edges best
// (smallest (max(left, right), min(left, right)), first one wins) and cuts every edge that crosses that
// line; the cut points become new vertices at the end of the table (convex, direction of the edge they
// sit on) and are what getObstacleVertex / getNextObstacleVertexNo (env.py:307-311) and the neighbour
// query see afterwards.  The kernels scan edges brute force, so only the cuts are kept, not the tree.
// Work list instead of recursion; a node's left set is expanded before its right set, which is the
// order in which the library numbers the new vertices.
static void split_crossing_edges(std::vector<ObstDev>& tab) {
    struct Side { int left, right; };
    auto sides = [&](int split, int edge, float* a, float* b) {
        const V2 s1 = mk(tab[split].px, tab[split].py), s2 = mk(tab[tab[split].next].px, tab[tab[split].next].py);
        *a = leftOf(s1, s2, mk(tab[edge].px, tab[edge].py));
        *b = leftOf(s1, s2, mk(tab[tab[edge].next].px, tab[tab[edge].next].py));
    };
    auto worse_or_equal = [](Side x, Side y) {  // (max, min) of x >= (max, min) of y
        const int xm = x.left > x.right ? x.left : x.right, xn = x.left < x.right ? x.left : x.right;
        const int ym = y.left > y.right ? y.left : y.right, yn = y.left < y.right ? y.left : y.right;
        return xm > ym || (xm == ym && xn >= yn);
    };
    std::vector<std::vector<int>> work(1);
    for (int i = 0; i < (int)tab.size(); ++i) work[0].push_back(i);
    while (!work.empty()) {
        const std::vector<int> edges = std::move(work.back());
        work.pop_back();
        const int n = (int)edges.size();
        if (n == 0) continue;
        int pick = 0;
        Side best = {n, n};
        for (int i = 0; i < n; ++i) {
            Side c = {0, 0};
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                float a, b;
                sides(edges[i], edges[j], &a, &b);
                if (a >= -EPS && b >= -EPS) ++c.left;
                else if (a <= EPS && b <= EPS) ++c.right;
                else { ++c.left; ++c.right; }
                if (worse_or_equal(c, best)) break;
            }
            if (!worse_or_equal(c, best)) { best = c; pick = i; }
        }
        std::vector<int> lhs, rhs;
        const int sp = edges[pick];
        const V2 s1 = mk(tab[sp].px, tab[sp].py), s2 = mk(tab[tab[sp].next].px, tab[tab[sp].next].py);
        for (int j = 0; j < n; ++j) {
            if (j == pick) continue;
            const int e1 = edges[j], e2 = tab[e1].next;
            float a, b;
            sides(sp, e1, &a, &b);
            if (a >= -EPS && b >= -EPS) { lhs.push_back(e1); continue; }
            if (a <= EPS && b <= EPS) { rhs.push_back(e1); continue; }
            const V2 p1 = mk(tab[e1].px, tab[e1].py), p2 = mk(tab[e2].px, tab[e2].py);
            const float t = det(s2 - s1, p1 - s1) / det(s2 - s1, p1 - p2);
            const V2 cut = p1 + t * (p2 - p1);
            ObstDev o;
            memset(&o, 0, sizeof o);
            o.px = cut.x; o.py = cut.y; o.ux = tab[e1].ux; o.uy = tab[e1].uy;
            o.prev = e1; o.next = e2; o.convex = 1;
            const int id = (int)tab.size();
            tab.push_back(o);
            tab[e1].next = id; tab[e2].prev = id;
            if (a > 0.0f) { lhs.push_back(e1); rhs.push_back(id); }
            else { rhs.push_back(e1); lhs.push_back(id); }
        }
        work.push_back(std::move(rhs));  // popped after the whole left subtree
        work.push_back(std::move(lhs));
    }
}

// Stream discipline: a handle's own stream is hipStreamNonBlocking, i.e. NOT ordered against the legacy
// null stream, and a fill of device memory through the blocking API is still asynchronous on the device
// (and split into a bulk and a tail command for sizes that are not a multiple of 8 bytes).  So nothing in
// this file touches the null stream: every fill and copy is the *Async form on the handle's stream, and
// an entry point that hands memory to later calls finishes with hipStreamSynchronize(e->stream).
template <class T>
static hipError_t dalloc(ca_env* e, T** p, size_t n, bool zero = true) {
    hipError_t r = hipMalloc((void**)p, n * sizeof(T));
    if (r != hipSuccess || !zero) return r;
    return hipMemsetAsync(*p, 0, n * sizeof(T), e->stream);
}
static hipError_t upload(ca_env* e, void* dst, const void* src, size_t bytes) {  // host -> device, complete on return
    hipError_t r = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream);
    return r != hipSuccess ? r : hipStreamSynchronize(e->stream);
}
static hipError_t download(ca_env* e, void* dst, const void* src, size_t bytes) {  // device -> host, complete on return
    hipError_t r = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream);
    return r != hipSuccess ? r : hipStreamSynchronize(e->stream);
}

// + the statically allocated LDS of the fused neighbour search: positions, and for >= 256 lanes the grid tables
static size_t lds_static_bytes(const ca_env* e) {
    return (e->help ? (size_t)e->KT * e->BS * 8 : 0) + (size_t)e->BS * 8 +
           (e->BS >= 1024 ? (size_t)e->BS * 2 + 16 + 1024 + 1028 : (e->BS >= 256 ? (size_t)e->BS * 2 + (size_t)e->BS * 8 + 16 + 4096 + 4100 : 0)) + 64;
}

// Which solve kernel (lane-per-agent family) serves this handle.  Register-resident ORCA lines (ST = 4 obstacle slots + KMAX
// neighbour slots, LP2 / LP1 unrolled, no scratch) when K <= 10 and
//   * the obstacle-neighbour list holds at most 4 entries (the synthetic crowds: one boundary polygon), or
//   * it holds up to 16 and the world is SMALL (at most 16 edges per arena: the reference env's own doorway world, 14 edges
//     after processObstacles, env.py:117-122; "congested", ALAN:195-208) -- there an agent practically never has more than
//     four edges in range (none in 3.8e5 sampled agent-steps of either world), and the one that does is solved apart, exactly
//     (ca_step.h solve_many_obstacles), so the list capacity stays RVO2's "every edge in range" --, or
//   * the LDS line table would not fit (arenas above 256 agents with many obstacle neighbours), or the batch would not be resident
//     with it (below).
// Else the LDS line table (ST = 0): worlds like the two-way tube ("deadlock": 42 edges, 17 % of the agent-steps with more
// than four in range).  Called by ca_create (no table yet) and again whenever tables are installed: every variant computes
// the same bits, so a handle may change variant between steps.
static void pick_variant(ca_env* e) {
    // (CA_REG_LINES, latched by ca_create: 0 forces the LDS line table, 1 the register lines wherever they exist)
    const bool allow = e->sw.reg_lines != 0, force = e->sw.reg_lines == 1;
    e->KT = e->K <= 5 ? 5 : (e->K <= 10 ? 10 : 16);
    bool table_fits;
    {
        const bool h0 = e->help;
        e->help = false;   // (the LDS line table never runs with helper lanes)
        table_fits = step_lds_bytes(e->BS, e->K, e->S, 0, e->KT) + lds_static_bytes(e) <= 160 * 1024;
        e->help = h0;
    }
    const bool small_world = e->max_edges <= 16;
    // ... or the batch is larger than the chip holds at once WITH the table: a 64-lane workgroup of the table kernel needs 28 KB of
    // LDS (K = 10, S = 16), five fit a CU, and from the 1281st workgroup on the launch runs in rounds of a kernel that is a third
    // occupied -- there the register lines win even in the two-way tube, where every sixth agent is solved apart (measured:
    // deadlock x 50 agents, 1280 arenas 49.6 us (table) / 64.8 us (registers) per ORCA step, 1536 arenas 81.6 / 67.7, 4096 arenas
    // 157.8 / 76.3; blocks x 20 agents, 8192 arenas 126.9 / 75.4; profiles/r04_g_many_edge_worlds.txt)
    bool table_resident = table_fits;
    if (table_fits) {
        const bool h0 = e->help;
        e->help = false;
        const size_t per_wg = step_lds_bytes(e->BS, e->K, e->S, 0, e->KT) + lds_static_bytes(e);
        e->help = h0;
        table_resident = (long)e->grid <= (long)e->n_cus * (long)((160 * 1024) / per_wg);
    }
    if (allow && e->K <= 10 && e->S <= 4) { e->ST = 4; e->SMX = 4; }
    else if (allow && e->K <= 10 && e->fuse_nbr && (small_world || force || !table_resident)) { e->ST = 4; e->SMX = 16; }
    else { e->ST = 0; e->SMX = 16; }
    e->lds = step_lds_bytes(e->BS, e->K, e->S, e->ST, e->KT);
    {   // helper lanes for the uniform-grid neighbour scan: arenas of 192 .. 512 agents on the register-line kernel
        e->help = e->sw.nbr_help != 0 && e->fuse_nbr   /* (CA_NBR_HELP=0: none) */ && e->ST > 0 && e->SMX == 4 && (e->BS == 256 || e->BS == 512) &&
                  e->cfg.n_agents >= 192 && e->K > 0;
    }
    {   // two lanes per agent for the whole step (ca_pair.h) where the helper lanes were: a 512-agent arena is 8 waves of one
        // lane per agent on its CU -- two per SIMD, each a long dependent chain; 16 waves with half the chain each fill it
        // (CA_PAIR=0: the lane kernel, with helper lanes in the scan from 192 agents; CA_PAIR_MIN: smallest arena that takes it --
        // arenas of 65 .. 128 agents, two waves, measured faster on the lane kernel: ca_pair.h)
        const int pair_min = e->sw.pair_min;
        e->pair = e->sw.pair != 0 && e->fuse_nbr && e->ST > 0 && e->SMX == 4 && (e->BS == 256 || e->BS == 512) &&
                  e->cfg.n_agents >= pair_min && e->K > 0;
        e->lds_p = pair_lds_bytes(e->BS, e->KT);
    }
}

// the dynamic-LDS limits of the kernels pick_variant chose -- and, for BOTH callers (ca_create, and install_tables whenever a
// world is installed and the variant may change), the check that the chosen kernel fits the CU's 160 KiB: the lane kernels by
// their dynamic part + the statically declared arrays of the fused neighbour search, the two-lanes kernel by what the
// compiled kernel itself reports (hipFuncGetAttributes).  *misfit = the kernel does not fit (the callers turn it into CA_ERANGE);
// the return value carries genuine HIP failures only.
static const size_t LDS_PER_CU = 160 * 1024;
static hipError_t apply_variant_attributes(ca_env* e, bool* misfit) {
    hipError_t r = hipSuccess;
    *misfit = false;
    if (!e->pair && e->lds + lds_static_bytes(e) > LDS_PER_CU) { *misfit = true; return hipSuccess; }
    if (r == hipSuccess && e->pair) {
        const void* f = e->KT == 5 ? (e->BS == 256 ? reinterpret_cast<const void*>(&pair_kernel<5, 256>)
                                                   : reinterpret_cast<const void*>(&pair_kernel<5, 512>))
                                   : (e->BS == 256 ? reinterpret_cast<const void*>(&pair_kernel<10, 256>)
                                                   : reinterpret_cast<const void*>(&pair_kernel<10, 512>));
        hipFuncAttributes fa;
        r = hipFuncGetAttributes(&fa, f);
        if (r == hipSuccess && fa.sharedSizeBytes + e->lds_p > LDS_PER_CU) { *misfit = true; return hipSuccess; }
        if (r == hipSuccess) r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_p);
    }
    if (r == hipSuccess && e->help && e->lds > 48 * 1024) {
        const void* f = e->KT == 5 ? (e->BS == 256 ? reinterpret_cast<const void*>(&step_kernel<5, 256, 4, true, 2>)
                                                   : reinterpret_cast<const void*>(&step_kernel<5, 512, 4, true, 2>))
                                   : (e->BS == 256 ? reinterpret_cast<const void*>(&step_kernel<10, 256, 4, true, 2>)
                                                   : reinterpret_cast<const void*>(&step_kernel<10, 512, 4, true, 2>));
        r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds);
    }
    if (r == hipSuccess && e->lds > 48 * 1024) {
        if (e->ST > 0) r = e->KT == 5 ? set_lds_attr_k<5, 4>(e->BS, e->lds) : set_lds_attr_k<10, 4>(e->BS, e->lds);
        else if (e->K <= 5) r = set_lds_attr_k<5, 0>(e->BS, e->lds);
        else if (e->K <= 10) r = set_lds_attr_k<10, 0>(e->BS, e->lds);
        else r = set_lds_attr_k<16, 0>(e->BS, e->lds);
    }
    return r;
}

// Which form the ALAN online step takes on this handle (ca_alan_configure, and again whenever pick_variant has run: the world decides
// the solve kernel).  One launch of the four-lanes kernel (alan_fused), one launch of a lane kernel of one or two waves (alan_lane:
// the softmax terms wait in the wave's LP3 pool -- 4 + KMAX doubles per lane -- or, LDS line table, in the table itself -- 2 (K + S)
// per lane), else select -> solve -> update.
static int alan_pick(ca_env* e) {
    e->alan_fused = e->alan_lane = false;
    if (e->n_actions <= 0) return CA_OK;
    const size_t lq = quad_lds_bytes(e->BSq, e->KT, e->SQ, e->n_actions);
    const bool on = e->sw.alan_fused != 0;   // (CA_ALAN_FUSED=0: the three-launch form everywhere)
    e->alan_fused = on && (e->quad || e->quad_roll) && lq <= 64 * 1024;
    e->alan_lane = on && !e->quad && e->fuse_nbr && !e->help && !e->pair && e->BS <= 128 && e->K <= 10 &&
                   e->n_actions <= (e->ST > 0 ? 4 + e->KT : 2 * (e->K + e->S));
    if (e->alan_fused && lq > 48 * 1024)
        HIPCHK(e, hipFuncSetAttribute(quad_fn(e, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lq));
    return CA_OK;
}

extern "C" {

const char* ca_last_error(const ca_env* env) { return env ? env->err.c_str() : g_create_err.c_str(); }

int ca_create(const ca_config* cfg, int device, void* stream, ca_env** out) {
    if (!cfg || !out) return fail(nullptr, CA_EINVAL, "ca_create: null argument");
    *out = nullptr;
    if (cfg->n_arenas <= 0 || cfg->n_agents <= 0 || cfg->n_agents > CA_MAX_AGENTS)
        return fail(nullptr, CA_ERANGE, "ca_create: n_arenas=%d n_agents=%d out of range (agents 1..%d)",
                    cfg->n_arenas, cfg->n_agents, CA_MAX_AGENTS);
    if (cfg->max_neighbors < 0 || cfg->max_neighbors > CA_MAX_NEIGHBORS)
        return fail(nullptr, CA_ERANGE, "ca_create: max_neighbors=%d out of range 0..%d", cfg->max_neighbors,
                    CA_MAX_NEIGHBORS);
    if (cfg->max_obst_neighbors < 1 || cfg->max_obst_neighbors > CA_MAX_OBST_NEIGHBORS)
        return fail(nullptr, CA_ERANGE, "ca_create: max_obst_neighbors=%d out of range 1..%d",
                    cfg->max_obst_neighbors, CA_MAX_OBST_NEIGHBORS);
    if (cfg->done_mode < 0 || cfg->done_mode > 2) return fail(nullptr, CA_EINVAL, "ca_create: bad done_mode");
    {   // the supported magnitudes (include/ca_env.h "Units and magnitudes"): the kernels' division and square root run without
        // their range-scaling instructions (ca_math.h) -- exact for worlds of O(1) units, not for metres-as-nanometres
        const struct { const char* name; float v, lo, hi; } rng[] = {
            {"time_step", cfg->time_step, CA_MIN_TIME_STEP, CA_MAX_TIME_STEP}, {"neighbor_dist", cfg->neighbor_dist, CA_MIN_LENGTH, CA_MAX_LENGTH},
            {"time_horizon", cfg->time_horizon, CA_MIN_LENGTH, CA_MAX_LENGTH}, {"time_horizon_obst", cfg->time_horizon_obst, CA_MIN_LENGTH, CA_MAX_LENGTH},
            {"radius", cfg->radius, CA_MIN_LENGTH, CA_MAX_LENGTH}, {"max_speed", cfg->max_speed, CA_MIN_LENGTH, CA_MAX_LENGTH}};
        for (const auto& q : rng)
            if (!(q.v >= q.lo && q.v <= q.hi))   // (also false for NaN)
                return fail(nullptr, CA_ERANGE, "ca_create: %s=%g outside the supported range [%g, %g]", q.name, (double)q.v, (double)q.lo, (double)q.hi);
        const float box[] = {cfg->spawn_x0, cfg->spawn_x1, cfg->spawn_y0, cfg->spawn_y1, cfg->goal_x0, cfg->goal_x1, cfg->goal_y0, cfg->goal_y1,
                             cfg->done_x_thresh};
        for (float v : box)
            if (!(fabsf(v) <= CA_MAX_COORD)) return fail(nullptr, CA_ERANGE, "ca_create: a spawn / goal box coordinate (%g) is beyond %g or not a number", (double)v, (double)CA_MAX_COORD);
    }
    if ((size_t)cfg->n_arenas * cfg->n_agents > (size_t)1 << 30)
        return fail(nullptr, CA_ERANGE, "ca_create: more than 2^30 agents on one handle");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, CA_ENODEV, "ca_create: no HIP device (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(nullptr, CA_ENODEV, "ca_create: device %d of %d", device, ndev);
    ca_env* e = new ca_env();
    e->cfg = *cfg;
    e->device = device;
    hipError_t r = hipSetDevice(device);
    if (r == hipSuccess) {
        if (stream) e->stream = (hipStream_t)stream;
        else { r = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking); e->own_stream = true; }
    }
    {   // the diagnostic switches, once
        auto digit = [](const char* name) { const char* v = getenv(name); return (v && v[0] >= '0' && v[0] <= '9') ? v[0] - '0' : -1; };
        e->sw.reg_lines = digit("CA_REG_LINES"); e->sw.nbr_help = digit("CA_NBR_HELP"); e->sw.pair = digit("CA_PAIR");
        e->sw.alan_fused = digit("CA_ALAN_FUSED");
        const char* pm = getenv("CA_PAIR_MIN");
        if (pm && atoi(pm) > 0) e->sw.pair_min = atoi(pm);
    }
    // launch geometry: P lanes per arena (power of two >= N), one or more whole arenas per block
    int P = 1, logP = 0;
    while (P < cfg->n_agents) { P <<= 1; ++logP; }
    e->P = P; e->logP = logP;
    e->BS = P > 64 ? P : 64;
    e->LS = P; e->apb = e->BS / P; e->linv = 65536 / P; e->dense = 0;   // (P a power of two: exact)
    e->grid = (cfg->n_arenas + e->apb - 1) / e->apb;
    {
        const char* v = getenv("CA_NBR_BS");  // diagnostic switch: power of two in [max(P, 64), 1024]
        const int want = v ? atoi(v) : 0;
        const bool ok = want >= 64 && want <= 1024 && (want & (want - 1)) == 0 && want >= P &&
                        CA_NBW16(want) == CA_NBW16(e->BS);  // the list entry width is a compile-time function of the block size
        e->BSn = ok ? want : e->BS;
        e->grid_n = (cfg->n_arenas + e->BSn / P - 1) / (e->BSn / P);
        const char* f = getenv("CA_FUSE_NBR");  // diagnostic switch: 0 = separate neighbour kernel
        e->fuse_nbr = !(f && f[0] == '0') && e->BSn == e->BS;
#ifndef CA_WITH_UNFUSED_NBR   // the product library has the fused form only (the stand-alone kernel: build with -DCA_WITH_UNFUSED_NBR)
        e->fuse_nbr = true; e->BSn = e->BS; e->grid_n = e->grid;
#endif
    }
    {
        const char* v = getenv("CA_OBS_DENSE");  // diagnostic switch: 0 = one arena per observation workgroup
        e->obs_dense_on = cfg->n_agents < 16 && !(v && v[0] == '0');
    }
    {   // arenas within one wave whose N is no power of two: N lanes each, back to back, where that packs more arenas into the
        // wave than P lanes each (the reference env's own 10-agent arenas: six per wave instead of four)
        const char* v = getenv("CA_DENSE");  // diagnostic switch: 0 = P lanes per arena
        const int N = cfg->n_agents;
        if (!(v && v[0] == '0') && e->fuse_nbr && e->BS == 64 && P < 64 && 64 / N > 64 / P) {
            const int linv = (65536 + N - 1) / N;
            bool exact = true;
            for (int t = 0; t < 64; ++t) exact = exact && ((t * linv) >> 16) == t / N;
            if (exact) {
                e->LS = N; e->apb = 64 / N; e->linv = linv; e->dense = 1;
                e->grid = (cfg->n_arenas + e->apb - 1) / e->apb;
            }
        }
    }
    e->K = cfg->max_neighbors;
    e->S = cfg->max_obst_neighbors;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->n_cus = cus;
    }
    pick_variant(e);
    {   // four lanes per agent (ca_quad.h) where one lane per agent would leave SIMDs without a wave: fewer than 1024
        // waves.  Measured crossover (profiles/r03_d_lane_vs_quad_by_batch_size.txt): 16-agent arenas -- quad ahead up to
        // 2048 arenas (512 lane-waves), behind from 4096 (1024); 64-agent arenas -- ahead up to 512 arenas, level at 1024.
        const char* v = getenv("CA_QUAD");  // 0 / 1 forces the choice (tests run the parity suite both ways)
        e->SQ = e->S <= 4 ? 4 : 16;
        // (register budget: a 1024-lane workgroup caps the kernel at 128 VGPRs, a 512-lane one at 256 -- the variant with 16
        // obstacle neighbours and K = 10 needs all 256 already at 256 lanes)
        const bool fits = e->K <= 10 && 4 * P <= ((e->SQ == 16 && e->KT == 10) ? 256 : 512);
        // (counted at P lanes per arena, the layout the crossover was measured with, whatever the packing is now)
        const long lane_waves = (long)((cfg->n_arenas + e->BS / P - 1) / (e->BS / P)) * (e->BS / 64);
        e->quad = fits && (v ? v[0] == '1' : lane_waves < 1024);
        e->quad_roll = fits && (v ? v[0] == '1' : lane_waves <= 1024);  // T steps per launch: ahead at 1024 lane-waves too
        e->BSq = 4 * P > 64 ? 4 * P : 64;
        const int apbq = (e->BSq / 4) / P;
        e->grid_q = (cfg->n_arenas + apbq - 1) / apbq;
        e->lds_q = quad_lds_bytes(e->BSq, e->KT, e->SQ);
    }
    if (!e->pair && e->lds + lds_static_bytes(e) > LDS_PER_CU) {
        fail(nullptr, CA_ERANGE, "ca_create: the solve kernel would need %zu B of LDS (> 160 KiB) for n_agents=%d, "
             "max_neighbors=%d, max_obst_neighbors=%d: arenas above 256 agents need max_neighbors <= 10 "
             "(register-line variant)", e->lds + lds_static_bytes(e), cfg->n_agents, e->K, e->S);
        if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
        delete e;
        return CA_ERANGE;
    }
    host_tables(e);
    const size_t an = AN(e), A = cfg->n_arenas;
    float** f32s[] = {&e->pos_x, &e->pos_y, &e->vel_x, &e->vel_y, &e->pref_x, &e->pref_y,
                      &e->tmp_x, &e->tmp_y, &e->orient_x, &e->orient_y};
    for (auto p : f32s) if (r == hipSuccess) r = dalloc(e, p, an);
    double** f64s[] = {&e->goal_x, &e->goal_y, &e->goal2_x, &e->goal2_y};
    for (auto p : f64s) if (r == hipSuccess) r = dalloc(e, p, an);
    int** i32s[] = {&e->agent_done, &e->arrive_step, &e->regoal_count};
    for (auto p : i32s) if (r == hipSuccess) r = dalloc(e, p, an);
    e->nidx16 = CA_NBW16(e->BS) ? 1 : 0;  // u8 indices address 256 agents (BS = max(64, pow2 >= N): the kernels' compile-time test)
    if (r == hipSuccess) r = dalloc(e, &e->counts, an);
    if (r == hipSuccess) r = dalloc(e, reinterpret_cast<unsigned char**>(&e->nb_idx),
                                    an * (size_t)(e->K > 0 ? e->K : 1) * (e->nidx16 ? 2 : 1));
    if (r == hipSuccess) r = dalloc(e, &e->obst_idx, an * (size_t)e->S);
    if (r == hipSuccess) r = dalloc(e, &e->slab, an * CA_OBS_DIM + an + 2 * A);   // observation | reward | arena_done | step_count
    if (r == hipSuccess) {
        e->obs = e->slab; e->reward = e->slab + an * CA_OBS_DIM;
        e->arena_done = reinterpret_cast<int*>(e->reward + an); e->step_count = e->arena_done + A;
    }
    if (r == hipSuccess) r = dalloc(e, &e->episode, A);
    if (r == hipSuccess) r = dalloc(e, &e->arena_stats, A * ST_STRIDE);
    if (r == hipSuccess) r = dalloc(e, &e->arena_steps, A);
    if (r == hipSuccess) r = dalloc(e, &e->d_obst, (size_t)1);
#ifdef CA_STAMPS
    if (r == hipSuccess) r = dalloc(e, &e->dbg, (size_t)std::max(std::max(e->grid * (2 * e->BS / 64), e->grid_n * (e->BSn / 64)), e->grid_q * (e->BSq / 64)) * 16);
    if (r == hipSuccess) r = dalloc(e, &e->dbg_obs, (size_t)cfg->n_arenas * ((cfg->n_agents + 15) / 16 + 16) * 4 * 16);
#endif
    bool lds_misfit = false;
    if (r == hipSuccess) r = apply_variant_attributes(e, &lds_misfit);
    if (r == hipSuccess && e->quad_roll && e->lds_q > 48 * 1024)
        r = hipFuncSetAttribute(quad_fn(e), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_q);
    if (r == hipSuccess) {
        const int obs_bs = obs_block_threads(cfg->n_agents);
        const size_t ol = obs_lds_bytes(obs_nstage(e), obs_bs, 16 * (e->K + e->S));
        if (ol > 48 * 1024) {
            r = hipFuncSetAttribute(reinterpret_cast<const void*>(obs_fn(obs_bs, e->nidx16 != 0, obs_dense(e))),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)ol);
        }
    }
    if (r == hipSuccess) {
        r = hipHostMalloc((void**)&e->ovf_host, sizeof(unsigned long long), hipHostMallocDefault);
        if (r == hipSuccess) *e->ovf_host = 0ull;
    }
    if (r == hipSuccess) r = hipMalloc((void**)&e->d_cold, sizeof(StepCold));
    if (r == hipSuccess) {
        StepCold hc;
        fill_cold(e, hc);
        r = upload(e, e->d_cold, &hc, sizeof hc);
    }
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);  // the zero fills are done before the handle is handed out
    if (r == hipSuccess && lds_misfit) {
        fail(nullptr, CA_ERANGE, "ca_create: the solve kernel for n_agents=%d, max_neighbors=%d, max_obst_neighbors=%d does not fit the "
             "160 KiB of LDS of a CU", cfg->n_agents, e->K, e->S);
        ca_destroy(e);
        return CA_ERANGE;
    }
    if (r != hipSuccess) {
        fail(nullptr, CA_EHIP, "ca_create: %s", hipGetErrorString(r));
        ca_destroy(e);
        return CA_EHIP;
    }
    *out = e;
    return CA_OK;
}

int ca_destroy(ca_env* e) {
    if (!e) return CA_OK;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    void* bufs[] = {e->pos_x, e->pos_y, e->vel_x, e->vel_y, e->pref_x, e->pref_y, e->goal_x, e->goal_y,
                    e->goal2_x, e->goal2_y, e->slab, e->tmp_x, e->tmp_y, e->orient_x, e->orient_y,
                    e->agent_done, e->arrive_step,
                    e->regoal_count, e->counts, e->nb_idx, e->obst_idx, e->cvt_buf, e->d_tab_off, e->d_cold, e->d_order,
                    e->episode, e->arena_stats, e->arena_steps, e->d_obst, e->dbg, e->dbg_obs,
                    e->alan_w, e->alan_t, e->alan_dirs, e->alan_u, e->alan_action, e->d_alan, e->mask_buf};
    for (void* b : bufs) if (b) hipFree(b);
    for (const auto& h : e->host_allocs) hipHostFree(h.first);
    if (e->ovf_host) hipHostFree(e->ovf_host);
    for (const ca_env::Span& sp : e->spans) { hipEventDestroy(sp.t0); hipEventDestroy(sp.t1); }
    for (hipEvent_t ev : e->free_events) hipEventDestroy(ev);
    if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
    delete e;
    return CA_OK;
}

int ca_set_stream(ca_env* e, void* stream) {
    if (!e) return CA_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->own_stream && e->stream) HIPCHK(e, hipStreamDestroy(e->stream));
    e->stream = (hipStream_t)stream;
    e->own_stream = false;
    return CA_OK;
}

// addObstacle for every polygon + processObstacles (env.py:118-123, 143-149): one processed edge table
static int build_table(ca_env* e, const float* verts_xy, const int32_t* poly_sizes, int n_poly, std::vector<ObstDev>& tab) {
    size_t off = 0;
    for (int pi = 0; pi < n_poly; ++pi) {
        const int n = poly_sizes[pi];
        if (n < 2) return fail(e, CA_EINVAL, "ca_set_obstacles: polygon %d has %d vertices (need >= 2)", pi, n);
        const int base = (int)tab.size();
        for (int i = 0; i < n; ++i) {  // SURVEY App. A.2 "Obstacle vertex attributes at addObstacle"
            const int in = (i == n - 1) ? 0 : i + 1, ip = (i == 0) ? n - 1 : i - 1;
            const V2 pt = mk(verts_xy[2 * (off + i)], verts_xy[2 * (off + i) + 1]);
            const V2 pn = mk(verts_xy[2 * (off + in)], verts_xy[2 * (off + in) + 1]);
            const V2 pp = mk(verts_xy[2 * (off + ip)], verts_xy[2 * (off + ip) + 1]);
            if (!(fabsf(pt.x) <= CA_MAX_COORD && fabsf(pt.y) <= CA_MAX_COORD))   // (also false for NaN)
                return fail(e, CA_ERANGE, "ca_set_obstacles: polygon %d vertex %d = (%g, %g) is beyond %g or not a number", pi, i,
                            (double)pt.x, (double)pt.y, (double)CA_MAX_COORD);
            if (pn.x == pt.x && pn.y == pt.y)
                return fail(e, CA_EINVAL, "ca_set_obstacles: polygon %d has an edge of length zero (vertex %d twice)", pi, i);
            if (absSq(pn - pt) < CA_MIN_EDGE * CA_MIN_EDGE)
                return fail(e, CA_ERANGE, "ca_set_obstacles: polygon %d edge %d is shorter than %g", pi, i, (double)CA_MIN_EDGE);
            const V2 u = normalize(pn - pt);
            ObstDev o;
            memset(&o, 0, sizeof o);
            o.px = pt.x; o.py = pt.y; o.ux = u.x; o.uy = u.y;
            o.next = base + in; o.prev = base + ip;
            o.convex = (n == 2) ? 1 : (leftOf(pp, pt, pn) >= 0.0f ? 1 : 0);
            tab.push_back(o);
        }
        off += n;
    }
    split_crossing_edges(tab);  // processObstacles (env.py:123)
    for (ObstDev& o : tab) {  // denormalise: each edge record also carries its two neighbours
        const ObstDev& nx = tab[o.next];
        const ObstDev& pv = tab[o.prev];
        o.qx = nx.px; o.qy = nx.py; o.qux = nx.ux; o.quy = nx.uy; o.qconvex = nx.convex;
        o.pux = pv.ux; o.puy = pv.uy;
    }
    return CA_OK;
}

// the obstacle-neighbour lists of the last step name edges of the OLD table(s): drop them (high byte of `counts`)
__global__ void clear_obst_counts_kernel(unsigned short* counts, size_t n) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) counts[q] &= 0x00FFu;
}

// installs the table(s): `all` = every table concatenated, `offs` empty (one table for all arenas) or [A + 1].
// The new tables are allocated and uploaded first and swapped in only when everything succeeded: a failure leaves the
// handle as it was.  The obstacle-neighbour lists left by the last step refer to the old tables, so their counts are
// cleared (an observation or reset before the next step then sees no obstacle segments instead of stale edge ids).
static int install_tables(ca_env* e, std::vector<ObstDev>& all, std::vector<int>& offs) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    ObstDev* n_obst = nullptr;
    int* n_off = nullptr;
    hipError_t r = hipMalloc((void**)&n_obst, (all.size() + 1) * sizeof(ObstDev));
    if (r == hipSuccess && !all.empty()) r = upload(e, n_obst, all.data(), all.size() * sizeof(ObstDev));
    if (r == hipSuccess && !offs.empty()) {
        r = hipMalloc((void**)&n_off, offs.size() * sizeof(int));
        if (r == hipSuccess) r = upload(e, n_off, offs.data(), offs.size() * sizeof(int));
    }
    if (r == hipSuccess) {
        const size_t an = AN(e);
        hipLaunchKernelGGL(clear_obst_counts_kernel, dim3((unsigned)((an + 255) / 256)), dim3(256), 0, e->stream, e->counts, an);
        r = hipGetLastError();
        if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    }
    if (r != hipSuccess) {
        if (n_obst) hipFree(n_obst);
        if (n_off) hipFree(n_off);
        return fail(e, CA_EHIP, "ca_set_obstacles: %s (the previous tables stay installed)", hipGetErrorString(r));
    }
    // the solve-kernel variant depends on the size of the world (pick_variant): the variant the NEW world selects is chosen and
    // checked against the CU's LDS while the old tables are still installed; a world whose kernel does not fit leaves the
    // handle exactly as it was (tables, variant, ALAN form)
    int me = 0;
    if (offs.empty()) me = (int)all.size();
    else for (size_t a = 0; a + 1 < offs.size(); ++a) me = std::max(me, offs[a + 1] - offs[a]);
    struct { int ST, SMX, KT, max_edges; bool help, pair; size_t lds, lds_p; } old_v = {e->ST, e->SMX, e->KT, e->max_edges, e->help, e->pair, e->lds, e->lds_p};
    e->max_edges = me;
    pick_variant(e);
    bool misfit = false;
    const hipError_t ra = apply_variant_attributes(e, &misfit);
    if (misfit || ra != hipSuccess) {
        e->ST = old_v.ST; e->SMX = old_v.SMX; e->KT = old_v.KT; e->max_edges = old_v.max_edges; e->help = old_v.help; e->pair = old_v.pair;
        e->lds = old_v.lds; e->lds_p = old_v.lds_p;
        bool dummy = false;
        (void)apply_variant_attributes(e, &dummy);   // (the previous variant's limits again: the attribute calls are idempotent)
        hipFree(n_obst);
        if (n_off) hipFree(n_off);
        if (misfit)
            return fail(e, CA_ERANGE, "ca_set_obstacles: the solve kernel this world selects does not fit the 160 KiB of LDS of a CU "
                        "(n_agents=%d, max_neighbors=%d, max_obst_neighbors=%d); the previous tables stay installed", e->cfg.n_agents, e->K, e->S);
        return fail(e, CA_EHIP, "ca_set_obstacles: %s (the previous tables stay installed)", hipGetErrorString(ra));
    }
    if (e->d_obst) hipFree(e->d_obst);
    if (e->d_tab_off) hipFree(e->d_tab_off);
    e->d_obst = n_obst;
    e->d_tab_off = n_off;
    e->h_obst.swap(all);
    e->h_tab_off.swap(offs);
    return alan_pick(e);   // (the form of the ALAN step follows the solve kernel)
}

int ca_set_obstacles(ca_env* e, const float* verts_xy, const int32_t* poly_sizes, int32_t n_poly) {
    if (!e || n_poly < 0 || (n_poly > 0 && (!verts_xy || !poly_sizes))) return fail(e, CA_EINVAL, "ca_set_obstacles: bad argument");
    std::vector<ObstDev> tab;
    const int rc = build_table(e, verts_xy, poly_sizes, n_poly, tab);
    if (rc) return rc;
    if (tab.size() > 65536) return fail(e, CA_ERANGE, "ca_set_obstacles: %zu edges (at most 65536)", tab.size());
    std::vector<int> none;
    return install_tables(e, tab, none);
}

int ca_set_obstacles_per_arena(ca_env* e, const float* verts_xy, const int32_t* poly_sizes, const int32_t* n_poly) {
    if (!e || !n_poly) return fail(e, CA_EINVAL, "ca_set_obstacles_per_arena: null argument");
    const int A = e->cfg.n_arenas;
    std::vector<ObstDev> all;
    std::vector<int> offs(1, 0);
    size_t voff = 0, poff = 0;
    for (int a = 0; a < A; ++a) {
        if (n_poly[a] < 0 || (n_poly[a] > 0 && (!verts_xy || !poly_sizes)))
            return fail(e, CA_EINVAL, "ca_set_obstacles_per_arena: bad argument for arena %d", a);
        std::vector<ObstDev> tab;
        const int rc = build_table(e, verts_xy ? verts_xy + 2 * voff : nullptr, poly_sizes ? poly_sizes + poff : nullptr,
                                   n_poly[a], tab);
        if (rc) return rc;
        for (int k = 0; k < n_poly[a]; ++k) voff += (size_t)poly_sizes[poff + k];
        poff += (size_t)n_poly[a];
        if (tab.size() > 65536) return fail(e, CA_ERANGE, "ca_set_obstacles_per_arena: arena %d has %zu edges (at most 65536)", a, tab.size());
        all.insert(all.end(), tab.begin(), tab.end());  // next / prev stay local to the arena's table
        if (all.size() > (size_t)0x7FFFFFFF) return fail(e, CA_ERANGE, "ca_set_obstacles_per_arena: too many edges");
        offs.push_back((int)all.size());
    }
    return install_tables(e, all, offs);
}

static int get_table(ca_env* e, int arena, float* verts_xy, int32_t* next, int32_t* convex, int32_t cap, int32_t* n_out) {
    const bool per = !e->h_tab_off.empty();
    const int t0 = per ? e->h_tab_off[arena] : 0;
    const int n = per ? e->h_tab_off[arena + 1] - t0 : (int)e->h_obst.size();
    *n_out = n;
    for (int i = 0; i < n && i < cap; ++i) {
        const ObstDev& o = e->h_obst[t0 + i];
        if (verts_xy) { verts_xy[2 * i] = o.px; verts_xy[2 * i + 1] = o.py; }
        if (next) next[i] = o.next;
        if (convex) convex[i] = o.convex;
    }
    return CA_OK;
}

int ca_get_obstacles(ca_env* e, float* verts_xy, int32_t* next, int32_t* convex, int32_t cap, int32_t* n_out) {
    if (!e || !n_out) return fail(e, CA_EINVAL, "ca_get_obstacles: null argument");
    return get_table(e, 0, verts_xy, next, convex, cap, n_out);
}

int ca_get_obstacles_arena(ca_env* e, int32_t arena, float* verts_xy, int32_t* next, int32_t* convex, int32_t cap,
                           int32_t* n_out) {
    if (!e || !n_out) return fail(e, CA_EINVAL, "ca_get_obstacles_arena: null argument");
    if (arena < 0 || arena >= e->cfg.n_arenas) return fail(e, CA_ERANGE, "ca_get_obstacles_arena: arena %d of %d", arena, e->cfg.n_arenas);
    return get_table(e, arena, verts_xy, next, convex, cap, n_out);
}

}  // extern "C" (the scenario generators below are plain C++)

// The scenario generators (env.py:86-97, ALAN:175-457) on the host, for arenas [0, An): An = n_arenas for the one scenario
// whose draws are sequential per arena (rejection-sampled starts), An = 1 for the per-agent TABLE of the others -- their
// layout either does not depend on the arena at all (circle, incoming, blocks, deadlock: libm's cos / sin / sqrt stay on the
// host) or only through counter-based draws, which init_scenario_kernel makes on the device.
static void host_scenario(const ca_env* e, int scenario, int An, std::vector<float>& px, std::vector<float>& py, std::vector<float>& vx,
                          std::vector<float>& vy, std::vector<float>& fx, std::vector<float>& fy, std::vector<double>& gx,
                          std::vector<double>& gy, std::vector<double>& g2x, std::vector<double>& g2y) {
    const ca_config& c = e->cfg;
    const int A = An, N = c.n_agents;
    const size_t an = (size_t)An * N;
    px.assign(an, 0.0f); py.assign(an, 0.0f); vx.assign(an, 0.0f); vy.assign(an, 0.0f); fx.assign(an, 0.0f); fy.assign(an, 0.0f);
    gx.assign(an, 0.0); gy.assign(an, 0.0); g2x.assign(an, 0.0); g2y.assign(an, 0.0);
    const double r = (double)c.radius;
    for (int a = 0; a < A; ++a) {
        const int64_t g = c.arena_offset + a;
        double theta = 0.0;
        for (int i = 0; i < N; ++i) {
            const size_t q = (size_t)a * N + i;
            double u0, u1, s, cs;
            rng2(c.seed, g, i, RNG_HEADING, 0, &u0, &u1);
            sincos64(uniform64(0.0, 2.0 * M_PI, u0), &s, &cs);  // env.py:89-90 / ALAN:276-277
            vx[q] = (float)cs; vy[q] = (float)s;
            if (scenario == CA_SCN_CROWD || scenario == CA_SCN_CROWD_SEPARATED) {  // ALAN:270-283
                const double E = std::sqrt(2.0 * r * N) * 2.0;
                // SURVEY 8d "rejection-sampled non-overlapping variant": the draw is repeated (sequence number
                // 1, 2, ...) until the start keeps 2 r from the starts of the agents before it; 64 tries at most
                const int tries = scenario == CA_SCN_CROWD_SEPARATED ? 64 : 1;
                const float minSq = sqr(c.radius + c.radius);
                for (int t = 0; t < tries; ++t) {
                    rng2(c.seed, g, i, RNG_POS, (uint32_t)t, &u0, &u1);
                    px[q] = (float)uniform64(0.0, E, u0); py[q] = (float)uniform64(0.0, E, u1);
                    bool clear = true;
                    for (int j = 0; j < i && clear; ++j)
                        clear = !(absSq(mk(px[q], py[q]) - mk(px[q - i + j], py[q - i + j])) < minSq);
                    if (clear) break;
                }
                rng2(c.seed, g, i, RNG_GOAL, 0, &u0, &u1);
                gx[q] = uniform64(0.0, E, u0); gy[q] = uniform64(0.0, E, u1);
                g2x[q] = gx[q]; g2y[q] = gy[q];
            } else if (scenario == CA_SCN_CIRCLE) {  // ALAN:297-322
                const double R = (r * 3 * N) / (2.0 * M_PI);
                const double E = 2.0 * R + 4.0 * r;
                px[q] = (float)(E / 2 + R * std::cos(theta)); py[q] = (float)(E / 2 + R * std::sin(theta));
                gx[q] = E / 2 + R * std::cos(theta + M_PI);
                gy[q] = E / 2 + R * std::sin(theta + M_PI);
                g2x[q] = gx[q]; g2y[q] = gy[q];
                theta += (2.0 * M_PI) / N;
            } else if (scenario >= CA_SCN_CONGESTED) {  // ALAN:175-193, 213-258, 333-357, 377-416
                const double E = (scenario == CA_SCN_CONGESTED) ? std::sqrt(2 * r * N) * 3
                               : (scenario == CA_SCN_BLOCKS) ? 3 * r * N : std::sqrt(2 * r * N) * 10;
                double x = 0, y = 0, tx = 0, ty = 0, t2x = 0, t2y = 0;
                if (scenario == CA_SCN_CONGESTED) {
                    rng2(c.seed, g, i, RNG_POS, 0, &u0, &u1);
                    x = uniform64(E * 0.2, E, u0); y = uniform64(0.0, E, u1);
                    tx = 0.1 * E - 1.0; ty = E / 2; t2x = 0.1 * E - E; t2y = E / 2;
                } else if (scenario == CA_SCN_INCOMING) {
                    if (i == 0) {
                        x = 0.1 * E; y = E / 2; tx = t2x = 0.9 * E; ty = t2y = E / 2;
                    } else {  // the block of N-1 agents, column by column (ALAN:232-258)
                        const double len = std::sqrt((double)(N - 1)), x_inc = 3 * r, y_inc = 2.1 * r;
                        const double y_start = E / 2 - ((y_inc * len) / 2);
                        double x_pos = 0.8 * E, y_pos = y_start;
                        for (int k = 1; k < i; ++k) {
                            y_pos += y_inc;
                            if (y_pos > y_start + y_inc * len) { x_pos += x_inc; y_pos = y_start; }
                        }
                        x = x_pos; y = y_pos; tx = t2x = x_pos - 0.7 * E; ty = t2y = y_pos;
                    }
                } else if (scenario == CA_SCN_BLOCKS) {
                    double y_pos = 1.5 * r;
                    for (int k = 0; k < i; ++k) y_pos += 3 * r;
                    x = 1.5 * r; y = y_pos; tx = t2x = E - 1.5 * r; ty = t2y = y_pos;
                } else {  // deadlock: two queues facing each other through a tube
                    const int half = N / 2;
                    if (i < half) {
                        double pos_x = 0.2 * E;
                        for (int k = 0; k < i; ++k) pos_x += -3 * r;
                        x = pos_x; tx = 0.9 * E; t2x = 0.9 * E + E;
                    } else {
                        double pos_x = 0.8 * E;
                        for (int k = half; k < i; ++k) pos_x += 3 * r;
                        x = pos_x; tx = 0.1 * E; t2x = 0.1 * E - E;
                    }
                    y = E / 2; ty = t2y = E / 2;
                }
                px[q] = (float)x; py[q] = (float)y; gx[q] = tx; gy[q] = ty;
                g2x[q] = t2x; g2y[q] = t2y;
            } else {  // env.py:86-95, 361
                const double E = 10.0;
                rng2(c.seed, g, i, RNG_POS, 0, &u0, &u1);
                px[q] = (float)uniform64(E * 0.5, E, u0); py[q] = (float)uniform64(0.0, E, u1);
                gx[q] = 1.0; gy[q] = 5.0; g2x[q] = -10.0; g2y[q] = 5.0;
            }
            double dx, dy;  // env.py:97 update_pref_vel
            pref_dir64(px[q], py[q], gx[q], gy[q], &dx, &dy);
            fx[q] = (float)dx; fy[q] = (float)dy;
        }
    }
}

struct InitArgs {
    float *pos_x, *pos_y, *vel_x, *vel_y, *pref_x, *pref_y;
    double *goal_x, *goal_y, *goal2_x, *goal2_y;
    const float *tpx, *tpy;                      // [N] table: start of agent i (arena-independent scenarios)
    const double *tgx, *tgy, *tg2x, *tg2y;       // [N] table: targets of agent i
    double x0, x1, y0, y1;                       // box of the drawn starts (pos_rng)
    double ge;                                   // drawn goals: uniform in (0, ge)^2 (goal_rng)
    uint64_t seed;
    int64_t arena_offset;
    int A, N, pos_rng, goal_rng;
};
// every agent of every arena: heading drawn (env.py:89-90 / ALAN:276-277), start and targets drawn or from the table,
// preferred velocity towards the target (env.py:97) -- the same counter-based streams and the same fp64 operations as the host
// path (ca_math.h), so the state is the host path's bit for bit
__global__ void init_scenario_kernel(const InitArgs p) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t)p.A * p.N) return;
    const int a = (int)(q / p.N), i = (int)(q - (size_t)a * p.N);
    const int64_t g = p.arena_offset + a;
    double u0, u1, s, cs;
    rng2(p.seed, g, i, RNG_HEADING, 0, &u0, &u1);
    sincos64(uniform64(0.0, 2.0 * M_PI, u0), &s, &cs);
    p.vel_x[q] = (float)cs; p.vel_y[q] = (float)s;
    float px = p.tpx[i], py = p.tpy[i];
    double gx = p.tgx[i], gy = p.tgy[i], g2x = p.tg2x[i], g2y = p.tg2y[i];
    if (p.pos_rng) {
        rng2(p.seed, g, i, RNG_POS, 0, &u0, &u1);
        px = (float)uniform64(p.x0, p.x1, u0); py = (float)uniform64(p.y0, p.y1, u1);
    }
    if (p.goal_rng) {
        rng2(p.seed, g, i, RNG_GOAL, 0, &u0, &u1);
        gx = uniform64(0.0, p.ge, u0); gy = uniform64(0.0, p.ge, u1);
        g2x = gx; g2y = gy;
    }
    double dx, dy;
    pref_dir64(px, py, gx, gy, &dx, &dy);
    p.pos_x[q] = px; p.pos_y[q] = py; p.pref_x[q] = (float)dx; p.pref_y[q] = (float)dy;
    p.goal_x[q] = gx; p.goal_y[q] = gy; p.goal2_x[q] = g2x; p.goal2_y[q] = g2y;
}

extern "C" {

int ca_init_scenario(ca_env* e, int32_t scenario) {
    if (!e) return CA_EINVAL;
    if (scenario < 0 || scenario > CA_SCN_CROWD_SEPARATED)
        return fail(e, CA_EINVAL, "ca_init_scenario: unknown scenario %d", scenario);
    const ca_config& c = e->cfg;
    const int A = c.n_arenas, N = c.n_agents;
    const size_t an = AN(e);
    const bool on_host = scenario == CA_SCN_CROWD_SEPARATED;   // sequential draws per arena (each start depends on the ones before)
    std::vector<float> px, py, vx, vy, fx, fy;
    std::vector<double> gx, gy, g2x, g2y;
    host_scenario(e, scenario, on_host ? A : 1, px, py, vx, vy, fx, fy, gx, gy, g2x, g2y);
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = e->stream;  // everything below is ordered on the handle's stream (see dalloc)
    if (on_host) {
        struct { float* d; std::vector<float>* h; } up[] = {{e->pos_x, &px}, {e->pos_y, &py}, {e->vel_x, &vx},
            {e->vel_y, &vy}, {e->pref_x, &fx}, {e->pref_y, &fy}};
        struct { double* d; std::vector<double>* h; } upd[] = {{e->goal_x, &gx}, {e->goal_y, &gy},
            {e->goal2_x, &g2x}, {e->goal2_y, &g2y}};
        for (auto& u : up) HIPCHK(e, hipMemcpyAsync(u.d, u.h->data(), an * 4, hipMemcpyHostToDevice, st));
        for (auto& u : upd) HIPCHK(e, hipMemcpyAsync(u.d, u.h->data(), an * 8, hipMemcpyHostToDevice, st));
    } else {   // the per-agent table (N entries) goes up, the kernel fills A x N agents
        char* tab = nullptr;
        const size_t nb = (size_t)N * (2 * 4 + 4 * 8);
        HIPCHK(e, hipMalloc((void**)&tab, nb));
        std::vector<char> h(nb);
        memcpy(h.data(), px.data(), (size_t)N * 4); memcpy(h.data() + (size_t)N * 4, py.data(), (size_t)N * 4);
        memcpy(h.data() + (size_t)N * 8, gx.data(), (size_t)N * 8); memcpy(h.data() + (size_t)N * 16, gy.data(), (size_t)N * 8);
        memcpy(h.data() + (size_t)N * 24, g2x.data(), (size_t)N * 8); memcpy(h.data() + (size_t)N * 32, g2y.data(), (size_t)N * 8);
        hipError_t r = hipMemcpyAsync(tab, h.data(), nb, hipMemcpyHostToDevice, st);
        if (r == hipSuccess) {
            InitArgs ia;
            ia.pos_x = e->pos_x; ia.pos_y = e->pos_y; ia.vel_x = e->vel_x; ia.vel_y = e->vel_y; ia.pref_x = e->pref_x; ia.pref_y = e->pref_y;
            ia.goal_x = e->goal_x; ia.goal_y = e->goal_y; ia.goal2_x = e->goal2_x; ia.goal2_y = e->goal2_y;
            ia.tpx = (const float*)tab; ia.tpy = (const float*)(tab + (size_t)N * 4);
            ia.tgx = (const double*)(tab + (size_t)N * 8); ia.tgy = (const double*)(tab + (size_t)N * 16);
            ia.tg2x = (const double*)(tab + (size_t)N * 24); ia.tg2y = (const double*)(tab + (size_t)N * 32);
            const double r2 = (double)c.radius;
            ia.pos_rng = (scenario == CA_SCN_CROWD || scenario == CA_SCN_CONGESTED || scenario == CA_SCN_DOORWAY) ? 1 : 0;
            ia.goal_rng = scenario == CA_SCN_CROWD ? 1 : 0;
            ia.x0 = 0.0; ia.x1 = 0.0; ia.y0 = 0.0; ia.y1 = 0.0; ia.ge = 0.0;
            if (scenario == CA_SCN_CROWD) { const double E = std::sqrt(2.0 * r2 * N) * 2.0; ia.x1 = E; ia.y1 = E; ia.ge = E; }             // ALAN:270-283
            else if (scenario == CA_SCN_CONGESTED) { const double E = std::sqrt(2 * r2 * N) * 3; ia.x0 = E * 0.2; ia.x1 = E; ia.y1 = E; }  // ALAN:177-186
            else if (scenario == CA_SCN_DOORWAY) { const double E = 10.0; ia.x0 = E * 0.5; ia.x1 = E; ia.y1 = E; }                        // env.py:86-95
            ia.seed = c.seed; ia.arena_offset = c.arena_offset; ia.A = A; ia.N = N;
            hipLaunchKernelGGL(init_scenario_kernel, dim3((unsigned)((an + 255) / 256)), dim3(256), 0, st, ia);
            r = hipGetLastError();
        }
        if (r == hipSuccess) r = hipStreamSynchronize(st);   // the table and the host staging go away
        hipFree(tab);
        if (r != hipSuccess) return fail(e, CA_EHIP, "ca_init_scenario: %s", hipGetErrorString(r));
    }
    HIPCHK(e, hipMemsetAsync(e->agent_done, 0, an * 4, st));
    HIPCHK(e, hipMemsetAsync(e->arrive_step, 0xff, an * 4, st));
    HIPCHK(e, hipMemsetAsync(e->regoal_count, 0, an * 4, st));
    HIPCHK(e, hipMemsetAsync(e->counts, 0, an * 2, st));
    HIPCHK(e, hipMemsetAsync(e->step_count, 0, (size_t)A * 4, st));
    HIPCHK(e, hipMemsetAsync(e->arena_done, 0, (size_t)A * 4, st));
    HIPCHK(e, hipMemsetAsync(e->episode, 0, (size_t)A * 4, st));
    HIPCHK(e, hipMemsetAsync(e->obs, 0, an * CA_OBS_DIM * 4, st));
    HIPCHK(e, hipStreamSynchronize(st));  // the host vectors above go out of scope
    e->orient_valid = false;
    return CA_OK;
}

// The packed neighbour-list fields keep their ABI image (i32 arrays, include/ca_env.h) through two small kernels.
__global__ void unpack_list_kernel(const void* src, int kind, int w16, int* dst, size_t n) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    if (kind == 0) dst[q] = (int)(reinterpret_cast<const unsigned short*>(src)[q] & 0xFFu);       // agent-neighbour count
    else if (kind == 1) dst[q] = (int)(reinterpret_cast<const unsigned short*>(src)[q] >> 8);      // obstacle-neighbour count
    else dst[q] = w16 ? ld_idx_t<true>(src, q) : ld_idx_t<false>(src, q);
}
__global__ void pack_list_kernel(const int* src, int kind, int w16, void* dst, size_t n) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    unsigned short* c = reinterpret_cast<unsigned short*>(dst);
    if (kind == 0) c[q] = (unsigned short)((c[q] & 0xFF00u) | ((unsigned)src[q] & 0xFFu));
    else if (kind == 1) c[q] = (unsigned short)((c[q] & 0x00FFu) | (((unsigned)src[q] & 0xFFu) << 8));
    else if (w16) st_idx_t<true>(dst, q, src[q]);
    else st_idx_t<false>(dst, q, src[q]);
}
static bool packed_field(int f) {
    return f == CA_FLD_NB_COUNT || f == CA_FLD_OBST_COUNT || f == CA_FLD_NB_IDX || f == CA_FLD_OBST_IDX;
}
static int packed_kind(int f) { return f == CA_FLD_NB_COUNT ? 0 : (f == CA_FLD_OBST_COUNT ? 1 : 2); }
static int cvt_reserve(ca_env* e, size_t elems) {
    if (e->cvt_cap >= elems) return CA_OK;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->cvt_buf) HIPCHK(e, hipFree(e->cvt_buf));
    e->cvt_buf = nullptr; e->cvt_cap = 0;
    HIPCHK(e, hipMalloc((void**)&e->cvt_buf, elems * sizeof(int)));
    e->cvt_cap = elems;
    return CA_OK;
}

int ca_set(ca_env* e, int32_t field, const void* src, size_t bytes, int32_t src_is_device) {
    if (!e || !src) return fail(e, CA_EINVAL, "ca_set: null argument");
    const FieldInfo fi = field_info(e, field);
    if (!fi.ptr) return fail(e, CA_EINVAL, "ca_set: unknown field %d", field);
    if (!fi.writable) return fail(e, CA_EINVAL, "ca_set: field %d is read-only", field);
    if (bytes != fi.bytes) return fail(e, CA_ESIZE, "ca_set: field %d holds %zu bytes, got %zu", field, fi.bytes, bytes);
    HIPCHK(e, hipSetDevice(e->device));
    if (packed_field(field)) {
        e->lists_trusted = false;   // (until the next solve launch has rewritten them)
        const size_t n = bytes / 4;
        const int rc = cvt_reserve(e, n);
        if (rc) return rc;
        HIPCHK(e, hipMemcpyAsync(e->cvt_buf, src, bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(pack_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->cvt_buf,
                           packed_kind(field), field == CA_FLD_NB_IDX ? e->nidx16 : 1, fi.ptr, n);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipStreamSynchronize(e->stream));  // the staging buffer is free again, the caller's host array too
        return CA_OK;
    }
    HIPCHK(e, hipMemcpyAsync(fi.ptr, src, bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
    if (!src_is_device) HIPCHK(e, hipStreamSynchronize(e->stream));
    if (field == CA_FLD_POS_X || field == CA_FLD_POS_Y || field == CA_FLD_GOAL_X || field == CA_FLD_GOAL_Y)
        e->orient_valid = false;
    return CA_OK;
}

int ca_get(ca_env* e, int32_t field, void* dst, size_t bytes, int32_t dst_is_device) {
    if (!e || !dst) return fail(e, CA_EINVAL, "ca_get: null argument");
    const FieldInfo fi = field_info(e, field);
    if (!fi.ptr) return fail(e, CA_EINVAL, "ca_get: unknown field %d", field);
    if (bytes != fi.bytes) return fail(e, CA_ESIZE, "ca_get: field %d holds %zu bytes, got %zu", field, fi.bytes, bytes);
    HIPCHK(e, hipSetDevice(e->device));
    if (packed_field(field)) {
        const size_t n = bytes / 4;
        const int rc = cvt_reserve(e, n);
        if (rc) return rc;
        hipLaunchKernelGGL(unpack_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, fi.ptr,
                           packed_kind(field), field == CA_FLD_NB_IDX ? e->nidx16 : 1, e->cvt_buf, n);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipMemcpyAsync(dst, e->cvt_buf, bytes, dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        return CA_OK;
    }
    HIPCHK(e, hipMemcpyAsync(dst, fi.ptr, bytes, dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream));
    if (!dst_is_device) HIPCHK(e, hipStreamSynchronize(e->stream));
    return CA_OK;
}

int ca_field_ptr(ca_env* e, int32_t field, void** dev_ptr, size_t* bytes) {
    if (!e || !dev_ptr) return fail(e, CA_EINVAL, "ca_field_ptr: null argument");
    const FieldInfo fi = field_info(e, field);
    if (!fi.ptr) return fail(e, CA_EINVAL, "ca_field_ptr: unknown field %d", field);
    if (packed_field(field))
        return fail(e, CA_EINVAL, "ca_field_ptr: field %d is stored packed (u8/u16); read it through ca_get", field);
    *dev_ptr = fi.ptr;
    if (bytes) *bytes = fi.bytes;
    return CA_OK;
}

int ca_bind_obs(ca_env* e, void* dev_ptr, size_t bytes) {
    if (!e || !dev_ptr) return fail(e, CA_EINVAL, "ca_bind_obs: null argument");
    if (bytes != AN(e) * CA_OBS_DIM * 4)
        return fail(e, CA_ESIZE, "ca_bind_obs: need %zu bytes, got %zu", AN(e) * CA_OBS_DIM * 4, bytes);
    if (((uintptr_t)dev_ptr & 15) != 0) return fail(e, CA_EINVAL, "ca_bind_obs: buffer must be 16-byte aligned");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->obs = (float*)dev_ptr;   // (the library's own observation buffer is part of the result slab and stays allocated)
    e->obs_external = true;
    return CA_OK;
}

int ca_reset(ca_env* e, const float* pos_x, const float* pos_y, int32_t pos_is_device, uint32_t flags) {
    if (!e) return CA_EINVAL;
    if ((pos_x == nullptr) != (pos_y == nullptr)) return fail(e, CA_EINVAL, "ca_reset: pos_x and pos_y go together");
    HIPCHK(e, hipSetDevice(e->device));
    StepArgs a;
    fill_args(e, a, nullptr, flags);
    if (pos_x) {
        if (pos_is_device) { a.reset_px = pos_x; a.reset_py = pos_y; }
        else {
            HIPCHK(e, hipMemcpyAsync(e->tmp_x, pos_x, AN(e) * 4, hipMemcpyHostToDevice, e->stream));
            HIPCHK(e, hipMemcpyAsync(e->tmp_y, pos_y, AN(e) * 4, hipMemcpyHostToDevice, e->stream));
            HIPCHK(e, hipStreamSynchronize(e->stream));  // the caller's host buffers may go away
            a.reset_px = e->tmp_x; a.reset_py = e->tmp_y;
        }
    }
    HIPCHK(e, launch_reset(e, a));
    HIPCHK(e, hipGetLastError());
    e->orient_valid = true;
    if (flags & CA_F_OBS) HIPCHK(e, launch_obs(e));
    return CA_OK;
}

int ca_reset_masked(ca_env* e, const int32_t* mask, int32_t mask_is_device, uint32_t flags) {
    if (!e || !mask) return fail(e, CA_EINVAL, "ca_reset_masked: null argument");
    HIPCHK(e, hipSetDevice(e->device));
    const int A = e->cfg.n_arenas;
    if (!mask_is_device) {
        if (!e->mask_buf) HIPCHK(e, dalloc(e, &e->mask_buf, (size_t)A, /*zero=*/false));  // fully overwritten below
        HIPCHK(e, upload(e, e->mask_buf, mask, (size_t)A * 4));
        mask = e->mask_buf;
    }
    StepArgs a;
    fill_args(e, a, nullptr, flags);
    a.reset_mask = mask;
    HIPCHK(e, launch_reset(e, a));
    HIPCHK(e, hipGetLastError());
    if (flags & CA_F_OBS) HIPCHK(e, launch_obs(e));
    return CA_OK;
}

static int do_step(ca_env* e, const float* actions, uint32_t flags) {
    { const int rs = overflow_status(e, "step"); if (rs) return rs; }
    if (e->prof_period > 1) e->profiling = (e->steps_done % (uint64_t)e->prof_period) == 0;
    StepArgs a;
    fill_args(e, a, actions, flags);
    HIPCHK(e, launch_step(e, a));
    e->orient_valid = true;
    if (flags & CA_F_OBS) HIPCHK(e, launch_obs(e));
    e->steps_done += 1;
    return CA_OK;
}

int ca_step(ca_env* e, const float* actions, uint32_t flags) {
    if (!e) return CA_EINVAL;
    if (!actions) return fail(e, CA_EINVAL, "ca_step: actions is null (use ca_orca_step for the ORCA-only step)");
    if (flags & CA_F_NODONE) return fail(e, CA_EINVAL, "ca_step: CA_F_NODONE applies to ca_orca_step only");
    HIPCHK(e, hipSetDevice(e->device));
    return do_step(e, actions, flags);
}

int ca_step_host(ca_env* e, const float* actions_host, uint32_t flags) {
    if (!e) return CA_EINVAL;
    if (!actions_host) return fail(e, CA_EINVAL, "ca_step_host: actions is null");
    if (flags & CA_F_NODONE) return fail(e, CA_EINVAL, "ca_step_host: CA_F_NODONE applies to ca_orca_step only");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpyAsync(e->tmp_x, actions_host, AN(e) * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));  // the caller's buffer may be reused right away
    return do_step(e, e->tmp_x, flags);
}

/* The host-array form of a step with ONE round trip: actions in (or NULL: the ORCA-only step), then observation | reward |
 * arena_done | step_count out in one copy, one synchronisation at the end.  The shape the reference's callers use -- one
 * environment per worker, results wanted on the host every step (run_rllib.py:77, 108; env.py:367-416) -- is latency:
 * ca_step_host + three ca_get cost four synchronisations.  Buffers from ca_host_alloc are page-locked AND device-visible: such
 * an action buffer is read by the kernel where it lies (no staging copy). */
int ca_step_packed(ca_env* e, const float* actions_host, uint32_t flags, void* out_host, size_t out_bytes) {
    if (!e || !out_host) return fail(e, CA_EINVAL, "ca_step_packed: null argument");
    const size_t an = AN(e), A = (size_t)e->cfg.n_arenas;
    const size_t obs_b = an * CA_OBS_DIM * 4, rest_b = an * 4 + 2 * A * 4;
    if (out_bytes != obs_b + rest_b) return fail(e, CA_ESIZE, "ca_step_packed: need %zu bytes, got %zu", obs_b + rest_b, out_bytes);
    if (actions_host && (flags & CA_F_NODONE)) return fail(e, CA_EINVAL, "ca_step_packed: CA_F_NODONE applies to the ORCA-only step");
    HIPCHK(e, hipSetDevice(e->device));
    const float* act = nullptr;
    if (actions_host) {
        bool mapped = false;
        for (const auto& h : e->host_allocs)   // a ca_host_alloc buffer holding the whole action array: device-visible as it is
            if ((const char*)actions_host >= (const char*)h.first && (const char*)actions_host + an * 4 <= (const char*)h.first + h.second) { mapped = true; break; }
        if (mapped) act = actions_host;
        else {
            HIPCHK(e, hipMemcpyAsync(e->tmp_x, actions_host, an * 4, hipMemcpyHostToDevice, e->stream));
            act = e->tmp_x;
        }
    }
    const int rc = do_step(e, act, flags);
    if (rc) return rc;
    char* out = (char*)out_host;
    if (!e->obs_external) {
        HIPCHK(e, hipMemcpyAsync(out, e->slab, obs_b + rest_b, hipMemcpyDeviceToHost, e->stream));
    } else {
        HIPCHK(e, hipMemcpyAsync(out, e->obs, obs_b, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipMemcpyAsync(out + obs_b, e->reward, rest_b, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));   // (also the point from which the caller may reuse its action buffer)
    return overflow_status(e, "ca_step_packed");       // (this very step's lists included: the stream is idle)
}

int ca_orca_step(ca_env* e, uint32_t flags) {
    if (!e) return CA_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    return do_step(e, nullptr, flags);
}

int ca_alan_configure(ca_env* e, const double* actions_xy, int32_t n_actions, double temp, double timewindow,
                      double time_step) {
    if (!e || !actions_xy) return fail(e, CA_EINVAL, "ca_alan_configure: null argument");
    if (n_actions < 1 || n_actions > CA_ALAN_MAX_ACTIONS)
        return fail(e, CA_ERANGE, "ca_alan_configure: n_actions=%d out of range 1..%d", n_actions, CA_ALAN_MAX_ACTIONS);
    if (!(temp > 0.0) || !(timewindow > 0.0) || !(time_step > 0.0))
        return fail(e, CA_EINVAL, "ca_alan_configure: temp, timewindow and time_step must be positive");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (double** b : {&e->alan_w, &e->alan_t, &e->alan_dirs, &e->alan_u}) { if (*b) hipFree(*b); *b = nullptr; }
    if (e->alan_action) { hipFree(e->alan_action); e->alan_action = nullptr; }
    e->n_actions = 0;
    const size_t an = AN(e);
    HIPCHK(e, dalloc(e, &e->alan_w, an * (size_t)n_actions));
    HIPCHK(e, dalloc(e, &e->alan_t, an * (size_t)n_actions));
    HIPCHK(e, dalloc(e, &e->alan_dirs, an * 4));
    HIPCHK(e, dalloc(e, &e->alan_u, an));
    HIPCHK(e, dalloc(e, &e->alan_action, an));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (int k = 0; k < n_actions; ++k) {  // (cos, sin) of atan2(y, x) = the normalised vector (ALAN:592-595)
        const double x = actions_xy[2 * k], y = actions_xy[2 * k + 1], len = std::sqrt(x * x + y * y);
        e->act_c[k] = len == 0.0 ? 1.0 : x / len;
        e->act_s[k] = len == 0.0 ? 0.0 : y / len;
    }
    e->n_actions = n_actions;
    e->alan_temp = temp; e->alan_window = timewindow; e->alan_dt = time_step;
    {   // the same arguments for the four-lanes kernel, which runs the bandit inside its launch (ca_quad.h)
        if (!e->d_alan) HIPCHK(e, hipMalloc((void**)&e->d_alan, sizeof(AlanCold)));
        AlanCold h;
        memset(&h, 0, sizeof h);
        h.w = e->alan_w; h.t = e->alan_t; h.action = e->alan_action; h.reward = e->reward;
        memcpy(h.act_c, e->act_c, sizeof h.act_c); memcpy(h.act_s, e->act_s, sizeof h.act_s);
        h.temp = temp; h.window = timewindow; h.dt = time_step; h.reward_scale = e->cfg.reward_scale; h.nA = n_actions;
        HIPCHK(e, upload(e, e->d_alan, &h, sizeof h));
    }
    return alan_pick(e);
}

int ca_alan_step(ca_env* e, const double* u, int32_t u_is_device, uint32_t flags) {
    if (!e) return CA_EINVAL;
    if (e->n_actions <= 0) return fail(e, CA_EINVAL, "ca_alan_step: call ca_alan_configure first");
    if (flags & (CA_F_AUTORESET | CA_F_NODONE))
        return fail(e, CA_EINVAL, "ca_alan_step: CA_F_AUTORESET / CA_F_NODONE do not apply (ALAN:106-123)");
    HIPCHK(e, hipSetDevice(e->device));
    { const int rs = overflow_status(e, "ca_alan_step"); if (rs) return rs; }
    if (u && !u_is_device) {  // stage the caller's host uniforms
        HIPCHK(e, hipMemcpyAsync(e->alan_u, u, AN(e) * 8, hipMemcpyHostToDevice, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        u = e->alan_u;
    }
    if (e->prof_period > 1) e->profiling = (e->steps_done % (uint64_t)e->prof_period) == 0;
    if ((e->alan_fused && e->quad) || e->alan_lane) {   // ONE launch: the bandit runs inside the solve kernel (ca_quad.h / ca_step.h)
        StepArgs a;
        fill_args(e, a, nullptr, flags);
        a.alan = e->d_alan; a.alan_u = u;
        HIPCHK(e, launch_step(e, a));
        e->orient_valid = true;
        if (flags & CA_F_OBS) HIPCHK(e, launch_obs(e));
        e->steps_done += 1;
        return CA_OK;
    }
    const ca_config& c = e->cfg;
    AlanArgs p;
    p.pos_x = e->pos_x; p.pos_y = e->pos_y; p.vel_x = e->vel_x; p.vel_y = e->vel_y;
    p.goal_x = e->goal_x; p.goal_y = e->goal_y; p.pref_x = e->pref_x; p.pref_y = e->pref_y; p.reward = e->reward;
    p.w = e->alan_w; p.t = e->alan_t; p.action = e->alan_action; p.dirs = e->alan_dirs; p.u = u;
    p.step_count = e->step_count; p.arena_done = e->arena_done; p.episode = e->episode; p.arena_stats = e->arena_stats;
    memcpy(p.act_c, e->act_c, sizeof p.act_c);
    memcpy(p.act_s, e->act_s, sizeof p.act_s);
    p.temp = e->alan_temp; p.window = e->alan_window; p.dt = e->alan_dt; p.reward_scale = c.reward_scale;
    p.seed = c.seed; p.arena_offset = c.arena_offset; p.A = c.n_arenas; p.N = c.n_agents; p.nA = e->n_actions;
    p.flags = flags;
    const dim3 grid((unsigned)((AN(e) + ALAN_BS - 1) / ALAN_BS)), block(ALAN_BS);
    {
        ProfScope ps(e, KIND_RESET);
        launch_k(ps, alan_select_kernel, grid, block, (size_t)e->n_actions * ALAN_BS * 8, e->stream, p);
    }
    HIPCHK(e, hipGetLastError());
    StepArgs a;
    fill_args(e, a, nullptr, flags);  // sim.doStep() + ALAN:118-121 = the ORCA-mode step
    HIPCHK(e, launch_step(e, a));
    {
        ProfScope ps(e, KIND_RESET);
        launch_k(ps, alan_update_kernel, grid, block, 0, e->stream, p);
    }
    HIPCHK(e, hipGetLastError());
    e->orient_valid = true;
    if (flags & CA_F_OBS) HIPCHK(e, launch_obs(e));
    e->steps_done += 1;
    return CA_OK;
}

int ca_alan_rollout(ca_env* e, int32_t steps, uint32_t flags) {
    if (!e || steps < 0) return fail(e, CA_EINVAL, "ca_alan_rollout: bad argument");
    { const int rs = overflow_status(e, "ca_alan_rollout"); if (rs) return rs; }
    if (e->n_actions > 0 && e->alan_fused && e->quad_roll && !(flags & (CA_F_OBS | CA_F_AUTORESET | CA_F_NODONE)) && steps > 1) {
        // run_sim(mode=1) (ALAN:106-123) as ONE launch per CA_ROLLOUT_MAX_T steps: select -> doStep -> update for every step
        // inside the four-lanes kernel, weights and times resident in LDS (ca_quad.h)
        HIPCHK(e, hipSetDevice(e->device));
        StepArgs a;
        fill_args(e, a, nullptr, flags);
        a.alan = e->d_alan;
        for (int done = 0; done < steps; done += CA_ROLLOUT_MAX_T) {
            if (e->prof_period > 1) e->profiling = (e->steps_done / (uint64_t)CA_ROLLOUT_MAX_T) % (uint64_t)e->prof_period == 0;
            a.T = steps - done < CA_ROLLOUT_MAX_T ? steps - done : CA_ROLLOUT_MAX_T;
            if (a.T == 1 && !e->quad) {   // (a lone last step is a step, not a rollout: the handle's step kernel takes it)
                const int rc = ca_alan_step(e, nullptr, 0, flags);
                if (rc) return rc;
                continue;
            }
            HIPCHK(e, launch_quad(e, a));
            e->lists_trusted = e->lists_trusted || !(flags & CA_F_FREEZE);
            e->steps_done += (uint64_t)a.T;
        }
        e->orient_valid = true;
        return CA_OK;
    }
    for (int s = 0; s < steps; ++s) {
        const int rc = ca_alan_step(e, nullptr, 0, flags);
        if (rc) return rc;
    }
    return CA_OK;
}

int ca_observe(ca_env* e) {
    if (!e) return CA_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, launch_obs(e));
    return CA_OK;
}

int ca_rollout(ca_env* e, int32_t steps, uint32_t flags) {
    if (!e || steps < 0) return fail(e, CA_EINVAL, "ca_rollout: bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    { const int rs = overflow_status(e, "ca_rollout"); if (rs) return rs; }
    if (e->quad_roll && !(flags & CA_F_OBS) && steps > 1) {
        // ONE launch per CA_ROLLOUT_MAX_T steps: the workgroup that owns an arena keeps it in registers / LDS for all of
        // them (ca_quad.h).  Bounded, so that a long rollout stays a sequence of kernels of a few milliseconds (a kernel
        // cannot be interrupted, and a watchdog may reset a GPU over one that runs for seconds).
        StepArgs a;
        fill_args(e, a, nullptr, flags);
        for (int done = 0; done < steps; done += CA_ROLLOUT_MAX_T) {
            if (e->prof_period > 1) e->profiling = (e->steps_done / (uint64_t)CA_ROLLOUT_MAX_T) % (uint64_t)e->prof_period == 0;
            a.T = steps - done < CA_ROLLOUT_MAX_T ? steps - done : CA_ROLLOUT_MAX_T;
            HIPCHK(e, launch_step(e, a));
            e->steps_done += (uint64_t)a.T;
        }
        e->orient_valid = true;
        return CA_OK;
    }
    for (int s = 0; s < steps; ++s) {
        const int rc = do_step(e, nullptr, flags);
        if (rc) return rc;
    }
    return CA_OK;
}

int ca_sync(ca_env* e) {
    if (!e) return CA_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return overflow_status(e, "ca_sync");
}

int ca_allow_obstacle_overflow(ca_env* e, int32_t allow) {
    if (!e) return CA_EINVAL;
    e->allow_overflow = allow != 0;
    return CA_OK;
}

int ca_get_stats(ca_env* e, ca_stats* out) {
    if (!e || !out) return fail(e, CA_EINVAL, "ca_get_stats: null argument");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t A = e->cfg.n_arenas;
    std::vector<unsigned long long> h(A * ST_STRIDE);
    HIPCHK(e, hipMemcpyAsync(h.data(), e->arena_stats, h.size() * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    ca_stats s;
    memset(&s, 0, sizeof s);
    for (size_t a = 0; a < A; ++a) {
        const unsigned long long* r = &h[a * ST_STRIDE];
        s.episodes += r[ST_EPISODES]; s.collisions += r[ST_COLL]; s.obst_collisions += r[ST_OBST_COLL];
        s.goals_reached += r[ST_GOALS]; s.obst_overflow += r[ST_OVERFLOW];
        double d;
        memcpy(&d, &r[ST_SUMREW], 8);
        s.sum_reward += d;
    }
    // agent-steps: counted by the kernels (every solve launch adds, per arena it advanced, the steps it advanced it by)
    std::vector<unsigned long long> hs(A);
    HIPCHK(e, hipMemcpyAsync(hs.data(), e->arena_steps, A * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    unsigned long long arena_steps = 0;
    for (size_t a = 0; a < A; ++a) arena_steps += hs[a];
    s.agent_steps = arena_steps * (uint64_t)e->cfg.n_agents;
    *out = s;
    return CA_OK;
}

int ca_reset_stats(ca_env* e) {
    if (!e) return CA_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemsetAsync(e->arena_stats, 0, (size_t)e->cfg.n_arenas * ST_STRIDE * 8, e->stream));
    HIPCHK(e, hipMemsetAsync(e->arena_steps, 0, (size_t)e->cfg.n_arenas * 8, e->stream));
    e->steps_done = 0;
    if (e->ovf_host && (*(volatile unsigned long long*)e->ovf_host >> 63)) {   // the sticky overflow status goes with the counters; a step
        HIPCHK(e, hipStreamSynchronize(e->stream));                              // still in flight must not set it again behind the clear
        *(volatile unsigned long long*)e->ovf_host = 0ull;
    }
    return CA_OK;
}

int ca_debug_math(ca_env* e, int32_t op, const void* in, void* out, int32_t n) {
    if (!e || !in || !out || n <= 0 || op < 0 || op > 7) return fail(e, CA_EINVAL, "ca_debug_math: bad argument");
    static const size_t in_b[] = {4, 8, 8, 16, 16, 8, 8, 4}, out_b[] = {4, 4, 16, 16, 16, 8, 4, 4};
    HIPCHK(e, hipSetDevice(e->device));
    void *di = nullptr, *dout = nullptr;
    HIPCHK(e, hipMalloc(&di, in_b[op] * n));
    HIPCHK(e, hipMalloc(&dout, out_b[op] * n));
    HIPCHK(e, upload(e, di, in, in_b[op] * n));
    hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, op, di, dout, n, e->cfg.seed);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, download(e, out, dout, out_b[op] * n));
    hipFree(di);
    hipFree(dout);
    return CA_OK;
}

#ifdef CA_STAMPS  // the two entry points below exist in the CA_STAMPS diagnostic build only (tools/stamps.py, tools/diag/placement.py)
/* Diagnostic (not declared in include/ca_env.h): install a block order for the solve kernel -- workgroup b then works on
 * the arenas of block order[b] (a permutation of 0 .. grid-1; host array) -- or remove it (NULL).  Results do not depend on it. */
int ca_debug_set_order(ca_env* e, const int32_t* order) {
    if (!e) return CA_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->d_order) { HIPCHK(e, hipFree(e->d_order)); e->d_order = nullptr; }
    if (!order) return CA_OK;
    if (!e->fuse_nbr) return fail(e, CA_EINVAL, "ca_debug_set_order: needs the fused neighbour search (CA_FUSE_NBR / CA_NBR_BS unset)");
    std::vector<char> seen((size_t)e->grid, 0);
    for (int b = 0; b < e->grid; ++b) {
        if (order[b] < 0 || order[b] >= e->grid || seen[order[b]]) return fail(e, CA_EINVAL, "ca_debug_set_order: not a permutation");
        seen[order[b]] = 1;
    }
    HIPCHK(e, hipMalloc((void**)&e->d_order, (size_t)e->grid * sizeof(int)));
    HIPCHK(e, upload(e, e->d_order, order, (size_t)e->grid * sizeof(int)));
    return CA_OK;
}

/* CA_STAMPS diagnostic build only: the pieces of the dispatch timeline of tools/diag/timeline.py.  ca_debug_clock launches a
 * one-wave kernel on the handle's stream that writes the device-wide 100 MHz counter at its start and end into slot `slot`
 * (in-stream fences around a dispatch: the kernel before it has ended, the kernel after it has not begun);
 * ca_debug_empty launches a kernel of the solve kernel's grid, block and LDS size whose waves only stamp their start and end. */
__global__ void debug_clock_kernel(unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (threadIdx.x == 0) out[0] = t0;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (threadIdx.x == 0) out[1] = t1;
}
__global__ void debug_empty_kernel(unsigned long long* dbg, int waves_per_block) {
    extern __shared__ float4 smem4[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (threadIdx.x == 4096) smem4[0] = make_float4(0.f, 0.f, 0.f, 0.f);   // (keeps the LDS allocation)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* r = dbg + ((size_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6)) * 16;
        r[12] = t0; r[11] = t1;
    }
}
static unsigned long long* g_dbg_clock = nullptr;
int ca_debug_clock(ca_env* e, int32_t slot) {
    if (!e || slot < 0 || slot >= 64) return fail(e, CA_EINVAL, "ca_debug_clock: bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    if (!g_dbg_clock) HIPCHK(e, hipMalloc((void**)&g_dbg_clock, 64 * 2 * 8));
    hipLaunchKernelGGL(debug_clock_kernel, dim3(1), dim3(64), 0, e->stream, g_dbg_clock + 2 * slot);
    return CA_OK;
}
int ca_debug_clock_read(ca_env* e, unsigned long long* out, int32_t n_slots) {
    if (!e || !g_dbg_clock || n_slots > 64) return fail(e, CA_EINVAL, "ca_debug_clock_read: bad argument");
    HIPCHK(e, download(e, out, g_dbg_clock, (size_t)n_slots * 2 * 8));
    return CA_OK;
}
int ca_debug_empty(ca_env* e) {
    if (!e || !e->dbg) return fail(e, CA_EINVAL, "ca_debug_empty: not a CA_STAMPS build");
    HIPCHK(e, hipSetDevice(e->device));
    ProfScope ps(e, KIND_RESET);   // (timed under "reset_kernels": the step launch in front of it keeps "step_kernel")
    launch_k(ps, debug_empty_kernel, dim3(e->grid), dim3(e->BS), (size_t)e->lds, e->stream, e->dbg, e->BS / 64);
    return CA_OK;
}

/* CA_STAMPS diagnostic build only (not declared in include/ca_env.h): per-wave phase time stamps
 * of the last step kernel, [waves][16] u64. */
int ca_debug_stamps(ca_env* e, unsigned long long* out, int32_t max_waves, int32_t* n_waves) {
    if (!e || !e->dbg) return fail(e, CA_EINVAL, "ca_debug_stamps: not a CA_STAMPS build");
    const bool obs = max_waves < 0;  // negative: the observation kernel's stamps
    if (obs) max_waves = -max_waves;
    const int nw = obs ? e->cfg.n_arenas * ((e->cfg.n_agents + 15) / 16) * 4 : (e->quad ? e->grid_q * (e->BSq / 64) : e->grid * ((e->pair ? 2 : 1) * e->BS / 64));
    if (n_waves) *n_waves = nw;
    const int n = nw < max_waves ? nw : max_waves;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, download(e, out, obs ? e->dbg_obs : e->dbg, (size_t)n * 16 * 8));
    return CA_OK;
}
#endif  // CA_STAMPS

/* Page-locked host memory for the results of the host-array calls (ca_get / ca_step_host into the same buffers every step:
 * the reference hands out the same dict objects every call, env.py:463-466).  A pageable destination takes the 67-MB
 * observation of 4096 x 64 agents at ~10 GB/s, a pinned one at the link's rate.  Owned by the handle. */
int ca_host_alloc(ca_env* e, size_t bytes, void** out) {
    if (!e || !out || bytes == 0) return fail(e, CA_EINVAL, "ca_host_alloc: bad argument");
    *out = nullptr;
    HIPCHK(e, hipSetDevice(e->device));
    void* p = nullptr;
    HIPCHK(e, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    e->host_allocs.push_back({p, bytes});
    *out = p;
    return CA_OK;
}
int ca_host_free(ca_env* e, void* p) {
    if (!e || !p) return fail(e, CA_EINVAL, "ca_host_free: bad argument");
    for (size_t k = 0; k < e->host_allocs.size(); ++k)
        if (e->host_allocs[k].first == p) {
            e->host_allocs.erase(e->host_allocs.begin() + k);
            HIPCHK(e, hipHostFree(p));
            return CA_OK;
        }
    return fail(e, CA_EINVAL, "ca_host_free: not a buffer of this handle");
}

int ca_profile(ca_env* e, int32_t period) {
    if (!e || period < 0) return fail(e, CA_EINVAL, "ca_profile: bad argument");
    e->prof_period = period;
    e->profiling = period == 1;
    if (period > 0) {  // the events of the first sampled steps exist before the caller's timed region starts
        HIPCHK(e, hipSetDevice(e->device));
        while (e->free_events.size() < 256) {
            hipEvent_t ev = nullptr;
            if (hipEventCreate(&ev) != hipSuccess) break;
            e->free_events.push_back(ev);
        }
    }
    return CA_OK;
}

int ca_profile_read(ca_env* e, int32_t counts[4], float mean_ms[4]) {
    if (!e || !counts || !mean_ms) return fail(e, CA_EINVAL, "ca_profile_read: null argument");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    double sum[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) counts[k] = 0;
    for (const ca_env::Span& sp : e->spans) {
        float ms = 0.0f;
        // (a launch that advanced T steps -- ca_rollout's one-launch form -- is reported per step, like the others)
        if (hipEventElapsedTime(&ms, sp.t0, sp.t1) == hipSuccess) { sum[sp.kind] += ms / (sp.steps > 0 ? sp.steps : 1); counts[sp.kind] += 1; }
        e->free_events.push_back(sp.t0);
        e->free_events.push_back(sp.t1);
    }
    e->spans.clear();
    for (int k = 0; k < 4; ++k) mean_ms[k] = counts[k] ? (float)(sum[k] / counts[k]) : 0.0f;
    return CA_OK;
}

int ca_launch_info(ca_env* e, int32_t* block, int32_t* grid, int32_t* lds_bytes, int32_t* obs_grid) {
    if (!e) return CA_EINVAL;
    if (block) *block = e->quad ? e->BSq : (e->pair ? 2 * e->BS : e->BS);
    if (grid) *grid = e->quad ? e->grid_q : e->grid;
    if (lds_bytes) *lds_bytes = (int32_t)(e->quad ? e->lds_q : (e->pair ? e->lds_p : e->lds));
    if (obs_grid) {
        const int apb = obs_block_threads(e->cfg.n_agents) / 16;
        *obs_grid = obs_dense(e) ? (int32_t)(((size_t)e->cfg.n_arenas * e->cfg.n_agents + apb - 1) / apb)
                                 : (int32_t)((size_t)e->cfg.n_arenas * ((e->cfg.n_agents + apb - 1) / apb));
    }
    return CA_OK;
}

#ifndef CA_SRC_SHA
#define CA_SRC_SHA "unknown"
#endif
const char* ca_source_sha(void) { return CA_SRC_SHA; }

int ca_solver_info(ca_env* e, int32_t* lanes_per_agent, int32_t* rollout_one_launch) {
    if (!e) return CA_EINVAL;
    if (lanes_per_agent) *lanes_per_agent = e->quad ? 4 : (e->pair ? 2 : 1);
    if (rollout_one_launch) *rollout_one_launch = e->quad_roll ? 1 : 0;
    return CA_OK;
}

}  // extern "C"
